/*
 * puresound_hip.h -- C ABI of libpuresound_hip.so: the MI355X (gfx950) implementation of the
 * PureSound separator forward path (encoder -> Conv-TasNet masker -> mask -> decoder -> clamp).
 *
 * The reference (mcw519/PureSound) is pure Python/PyTorch and has no FFI; the seam a maintainer
 * would bind is the nn.Module contract.  Each entry point below names the reference method whose
 * arithmetic it replaces (file:line under the reference tree).  INTEGRATION.md shows the ctypes
 * stub that binds them from the files under puresound/nnet/.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - tensors are fp32; activations use the padded layout [N][C][ldt] with ldt % 128 == 0,
 *     ldt >= T, row-major, frames contiguous ("frame-major rows");
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work,
 *     they never allocate, synchronise or copy to the host, so they can be captured in a hipGraph;
 *   - return value: 0 on success, a negative PS_E_* code on argument errors, a positive hipError_t
 *     on launch failure; ps_last_error() gives a thread-local message;
 *   - inputs are never modified unless a parameter is documented in/out.
 */
#ifndef PURESOUND_HIP_H
#define PURESOUND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_ABI_VERSION 21

#define PS_E_INVALID (-1)     /* bad shape / null pointer / unsupported combination */
#define PS_E_ALIGN (-2)       /* ldt or a pointer violates the alignment contract */
#define PS_E_UNSUPPORTED (-3) /* valid in the reference, not implemented here (says which) */

/* normalisation that follows a convolution; applied in the CONSUMER's prologue */
#define PS_NORM_NONE 0   /* identity */
#define PS_NORM_GLOBAL 1 /* gLN / gGN: per-utterance mean,var over [C,T] (lobe/norm.py:20-34,96) */
#define PS_NORM_AFFINE 2 /* per-channel scale/shift only (eval BatchNorm1d folded; lobe/norm.py:94) */

#define PS_ACT_LINEAR 0
#define PS_ACT_RELU 1
#define PS_ACT_SIGMOID 2

#define PS_OUT_CLAMP 0   /* clamp_(-1,1)  (base_nn.py:415-416) */
#define PS_OUT_SIGMOID 1 /* sigmoid       (base_nn.py:418-419) */
#define PS_OUT_NONE 2    /* raw decoder output (module-level FreeEncDec.inverse) */

int ps_abi_version(void);
const char* ps_last_error(void);

/* Optional launch timing (bench.py's roofline leg): while enabled every kernel launch made by this
 * library is bracketed by hipEvents on the launch stream.  ps_profile_enable(1) clears old records.
 * ps_profile_read synchronises the recorded events and returns the summed duration and launch count
 * of one kernel family ("conv1x1", "dwconv", "free_encode", "free_decode", "embed_bias", "pad_rows",
 * "unpad_rows", "frame", "complex_mask", "istft_ola", "attn_stats_pool", "lstm", "lstm_cell", "chan_layernorm", "unfold_taps", "gated_product", "segment_overlap",
 * "film_conv", "lstm_gates_cell", "proj_layernorm", "overlap_average", "stream_windows", "stream_overlap", "conv1x1_bf16", "unfold2d", "conv2d", "activation", "add", "magnitude", "real_mask", "norm_activation", "self_attention", "add_position",
 * "film_apply").  Not for use under stream capture. */
int ps_debug_flags(int flags); /* test/profiling hooks; bits 0..7, 20..30: kernel-variant switches (named where they are
                                  tested); bits 8..19: cap of the conv1x1 persistent grid (0 = off); <0 reads; returns
                                  the old value */
int ps_debug_buffer(void* device_buffer); /* 6 x u64 per conv1x1 workgroup: s_memtime stamps + HW ids */
int ps_profile_enable(int on);
int ps_profile_read(const char* kernel, double* total_ms, int* launches);

/* Number of doubles a stats buffer needs: partial (sum, sum-of-squares) slabs per utterance.
 * Stats are handed from the producing kernel to the consuming kernel as per-workgroup partials
 * (deterministic; no atomics).  ps_stats_parts() is an upper bound for any kernel here. */
int ps_stats_parts(int channels, int frames);
/* exact number of partial slabs per utterance each producer writes ([N][parts][2] doubles) */
int ps_conv1x1_stats_parts(int M, int T);
int ps_dwconv_stats_parts(int H, int T);

/* round up T to the padded row length the kernels require */
int ps_padded_frames(int frames);

/* ---------------------------------------------------------------------------------------------
 * Layout helpers: compact [N][C][T] <-> padded [N][C][ldt].
 * ------------------------------------------------------------------------------------------- */
int ps_pad_rows_f32(const float* src, float* dst, int64_t rows, int T, int ldt, void* stream);
int ps_unpad_rows_f32(const float* src, float* dst, int64_t rows, int T, int ldt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FreeEncDec.forward (puresound/nnet/lobe/encoder.py:71-83):
 *   feats[n][c][t] = act( sum_j w[c][j] * wav[n][t*hop + j] ),  T = floor((L-win)/hop)+1
 * w is the checkpoint tensor encoder.weight [C][1][win] as stored.  relu = output_active.
 * ------------------------------------------------------------------------------------------- */
int ps_free_encode_f32(const float* wav, const float* w, float* feats, int N, int L, int C, int win,
                       int hop, int T, int ldt, int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * get_mask + apply_tf_masks(real,real) + FreeEncDec.inverse + _wav_output_constrain
 * (base_nn.py:81-95, 146-159, 414-424; lobe/encoder.py:85-94):
 *   e = feats * act(mask)            (mask == NULL: e = feats)
 *   out[n][s] = constrain( sum_c sum_{t: 0 <= s-t*hop < win} w[c][s-t*hop] * e[n][c][t] )
 * w is decoder.weight [C][1][win].  out is compact [N][Lout], Lout = (T-1)*hop+win.
 * ------------------------------------------------------------------------------------------- */
int ps_free_decode_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out,
                       int N, int C, int T, int ldt, int win, int hop, int out_mode, void* stream);
/* The same with a scratch buffer (ps_free_decode_workspace_bytes; 0 = this shape has no use for one): for win = 32,
 * hop = 16 on long rows the decoder then runs on the matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 products; a tile of
 * 32 frames per wave, the tiles' boundary samples completed by a second small launch).  Without a workspace, or for any
 * other shape, it is ps_free_decode_f32. */
size_t ps_free_decode_workspace_bytes(int N, int T, int win, int hop);
int ps_free_decode_ws_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out, int N, int C,
                          int T, int ldt, int win, int hop, int out_mode, void* workspace, size_t workspace_bytes,
                          void* stream);
/* The decoder with the signal score's moments as its epilogue (_align_waveform, base_nn.py:398-412, then
 * SDRLoss.forward / si_snr, loss/sdr.py:104-183, 263-299 -- all algebra on five moments, see ps_wave_moments_f64 below).
 * Writes `out` as ps_free_decode_ws_f32 does and, per utterance, P = ps_free_decode_moments_parts(...) slots of
 * (sum a, sum b, sum a^2, sum b^2, sum ab) in fp64 into partials [N][P][5], a = the estimate after the output constraint,
 * b = the aligned reference: ref [N][ldr] holds ref_len samples per row; shorter than the (T-1) hop + win output samples it
 * counts as left-padded with zeros, longer it is cut (the reference's rule).  The caller adds the P slots up in order
 * (deterministic: one writer per slot).  ps_free_decode_moments_parts = 0 where only the two-launch form exists
 * (anything but win = 32, hop = 16, T >= 64, C % 16 == 0): ps_free_decode_moments_f32 then returns PS_E_UNSUPPORTED. */
int ps_free_decode_moments_parts(int N, int C, int T, int ldt, int win, int hop);
int ps_free_decode_moments_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out, int N,
                               int C, int T, int ldt, int win, int hop, int out_mode, const float* ref, int ldr,
                               int ref_len, double* partials, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv-STFT encoder / iSTFT decoder (ConvSTFT.forward / .inverse, lobe/encoder.py:358-456; extend_fbins,
 * overlap_add, torch_window_sumsquare, lobe/stft.py:103-125).  wsin/wcos are trainable in every recipe, so both
 * transforms are dense products on ps_conv1x1_f32 (analysis: K = n_fft window samples; synthesis: K = 2*bins
 * with the Hermitian extension folded into the weight); these entry points are the steps around them:
 *   ps_frame_f32        frames[n][k][t] = wav[n][t*hop + k]                     (conv1d's stride)
 *   ps_complex_mask_f32 complex product of [re;im] channel halves with act(mask) (base_nn.py:56-61,97-112)
 *   ps_istft_ola_f32    out = constrain( OLA-sum(frames * window / n_fft) / OLA-sum(window^2) where > 1e-10 )
 * ------------------------------------------------------------------------------------------- */
int ps_frame_f32(const float* wav, float* frames, int N, int L, int win, int hop, int T, int ldt, void* stream);
int ps_complex_mask_f32(const float* feats, const float* mask, float* out, int N, int half, int ldt,
                        int mask_act, void* stream);
int ps_istft_ola_f32(const float* frames, const float* window, float* out, int N, int n_fft, int hop, int T,
                     int ldt, int out_mode, void* stream);
/* ps_polar_mask_f32: _apply_complex_mask_on_polar (base_nn.py:161-190) on [re;im] channel halves: magnitudes
 * multiply (the mask's through tanh), phases add, back to [re;im].
 * ps_magphase_f32: the conv-STFT's "MagPhase" output (lobe/encoder.py:384-389) from the analysis product
 * [re ; im = -conv(x, wsin)]: rows [mags ; phase], mags = re^2 + im^2 (sqrt(. + 1e-8) when take_sqrt: trainable
 * kernels), phase = atan2(im + 0.0, re). */
int ps_polar_mask_f32(const float* feats, const float* mask, float* out, int N, int half, int ldt, void* stream);
int ps_magphase_f32(const float* spec, float* out, int N, int half, int ldt, int take_sqrt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused 1x1 convolution (the Conv-TasNet channel-mixing GEMM; exact-fp32 MFMA):
 *   a[k][t]   = prologue(x[n][k][t])                       -- norm-apply + PReLU of the PRODUCER
 *   y[n][m][t] = sum_k W[m][k] * a[k][t] + bias[m] (+ bias_n[n][m]) (+ res[n][m][t])
 *   ostats[n][part] = partial (sum, sumsq) of y over valid (m < M, t < T)
 * Replaces: TCN.in_conv conv (conv_tasnet.py:43-46), DepthwiseSeparableConv1d.pointwise conv
 * (lobe/cnn.py:75-79), TCN.out_conv + residual (conv_tasnet.py:65,87-88), together with the
 * GlobLN / GroupNorm(1) / BatchNorm1d(eval) + PReLU that precede them (lobe/norm.py:20-34,94,96).
 *
 * wt is the weight in kernel layout [ceil(M/256)][Kp][256], Kp = ceil16(K): transposed (k-major), zero
 * padded, one 256-channel output tile after the other.
 * ------------------------------------------------------------------------------------------- */
typedef struct ps_prologue {
  int norm;              /* PS_NORM_* */
  int prelu;             /* 1: apply PReLU(slope[0]) after the norm */
  const double* stats;   /* [N][parts][2] partials written by the producer (PS_NORM_GLOBAL) */
  int parts;
  double count;          /* number of elements the statistics cover (C*T of the producer) */
  float eps;
  const float* gamma;    /* [K] gain  (PS_NORM_GLOBAL) or folded scale (PS_NORM_AFFINE) */
  const float* beta;     /* [K] bias  (PS_NORM_GLOBAL) or folded shift (PS_NORM_AFFINE) */
  const float* slope;    /* [1] PReLU slope */
  int pre_relu;          /* 1: ReLU BEFORE the norm (AttentiveStatisticsPooling.tdnn: Conv -> ReLU -> BatchNorm) */
  int post_tanh;         /* 1: tanh after the norm (pooling.py:82,108) instead of / in addition to PReLU */
} ps_prologue;

int ps_conv1x1_f32(const float* x, const float* wt, float* y, int N, int K, int M, int T, int ldt,
                   const ps_prologue* pro, const float* bias, const float* bias_n /* [N][M] or NULL */,
                   const float* res /* [N][M][ldt] or NULL; may alias y */, double* ostats /* or NULL */,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused depthwise dilated convolution (lobe/cnn.py:62-74, 97):
 *   a = prologue(x);  y[n][h][t] = b[h] + sum_j w[h][j] * a[n][h][t + j*dilation - left]  (0 outside [0,T))
 * left = ((P-1)/2)*dilation non-causal, (P-1)*dilation causal (the reference pads both sides and
 * trims the tail after the pointwise conv, lobe/cnn.py:58-60,100-101 -- same result).
 * w is depthwise.0.weight [H][1][P].
 * ------------------------------------------------------------------------------------------- */
int ps_dwconv_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int T, int ldt,
                  int P, int dilation, int left, const ps_prologue* pro, double* ostats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-utterance embedding bias for TCN blocks built with emb_dim > 0 (conv_tasnet.py:80-85):
 * the reference repeats dvec over T and concatenates it to x; the constant channels contribute
 *   bias_n[n][m] = sum_e W[m][C+e] * dvec_hat[n][e],  dvec_hat = dvec / max(||dvec||_2, 1e-12) if normalize
 * (F.normalize, conv_tasnet.py:348-349).  w_embed is [M][E] (the trailing E columns of in_conv.0.weight).
 * ------------------------------------------------------------------------------------------- */
int ps_embed_bias_f32(const float* dvec, const float* w_embed, float* bias_n, int N, int E, int M,
                      int normalize, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attentive statistics pooling tail (AttentiveStatisticsPooling.forward with lengths=None,
 * lobe/pooling.py:109-126): per (utterance, channel) softmax of the attention logits over frames, then the
 * attention-weighted mean and standard deviation of x:
 *   a = softmax_t(logits[n][c][:T]);  mean = sum a*x;  std = sqrt(max(sum a*(x-mean)^2, eps))
 *   out[n][c] = mean, out[n][C + c] = std          (the reference's cat((mean, std), 1).unsqueeze(2))
 * ------------------------------------------------------------------------------------------- */
int ps_attn_stats_pool_f32(const float* logits, const float* x, float* out, int N, int C, int T, int ldt,
                           float eps, void* stream);
/* The same with the `lengths` argument of the reference (lobe/pooling.py:87-107): lengths is [N] relative lengths in
 * (0, 1] (NULL = all ones); frame t of utterance n takes part iff (float)t < lengths[n] * (float)T, which is
 * length_to_mask(lengths * L) (pooling.py:9-50) followed by masked_fill(-inf). */
int ps_attn_stats_pool_len_f32(const float* logits, const float* x, const float* lengths, float* out, int N, int C,
                               int T, int ldt, float eps, void* stream);
/* the attention map itself (AttentiveStatisticsPooling.forward(..., return_weight=True), lobe/pooling.py:109-113):
 * out[n][c][t] = softmax over the valid frames of logits[n][c][:], 0 on masked frames; lengths as above or NULL */
int ps_attn_weights_f32(const float* logits, const float* lengths, float* out, int N, int C, int T, int ldt,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * ps_conv1x1_bf16_f32: the same operation as ps_conv1x1_f32 (same prologue, bias, residual, statistics, fp32 tensors
 * in HBM, fp32 accumulation) with the products on the bf16 matrix pipe.
 *   planes = 1: operands rounded to bf16 -- the arithmetic BASELINE.json names for its bf16 configurations.
 *   planes = 3: fp32-accurate: each operand is the sum of three bf16 terms and the six products down to 2^-16 of the
 *               leading one are accumulated (dropped terms <= 2^-24 relative).
 * wt_planes: the weight split the same way and laid out [ceil(M/256)][ceil(K/16)][planes][256][16] bf16 (zero
 * padded); ps_conv1x1_bf16_weight_bytes gives its size.  K <= 512 when a prologue transform is present.
 * ------------------------------------------------------------------------------------------- */
size_t ps_conv1x1_bf16_weight_bytes(int M, int K, int planes);
int ps_conv1x1_bf16_f32(const float* x, const void* wt_planes, float* y, int N, int K, int M, int T, int ldt, int planes,
                        const ps_prologue* pro, const float* bias, const float* bias_n, const float* res,
                        double* ostats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GatedTCN (conv_tasnet.py:129-215): its dense dilated convolutions run as ps_conv1x1_f32 over unfolded taps.
 *
 * ps_unfold_taps_f32: y[n][j*(K+E) + k][t] = x'[n][k][t + j*dilation - left] (0 outside [0,T)), where
 *   x' = scale[n][k] * x + shift[n][k] when scale/shift are given (FiLM conditioning, conv_tasnet.py:196-201) and
 *   rows k >= K are the constant embed[n][k-K] (the concatenated, repeated embedding, conv_tasnet.py:186-191; it
 *   is zero-padded by the convolution like any other channel).  The conv weight W[m][k][j] is then the matrix
 *   Wu[m][j*(K+E)+k].
 * ps_gated_product_f32: y = PReLU(norm(left)) * sigmoid(PReLU(norm(right))) with gLN (partial statistics from the
 *   producing ps_conv1x1_f32) or folded bN1d prologues (conv_tasnet.py:203-206); cLN goes through
 *   ps_chan_layernorm_f32 instead.
 * ------------------------------------------------------------------------------------------- */
int ps_unfold_taps_f32(const float* x, float* y, int N, int K, int T, int ldt, int P, int dilation, int left,
                       const float* scale, const float* shift, const float* embed, int E, void* stream);
/* the same with T_out >= T output frames per row (inputs beyond T read as zero): the reference's causal gated block pads
 * both sides and trims only after its output conv, so its norms see T + padding frames (conv_tasnet.py:203-211) */
int ps_unfold_taps_out_f32(const float* x, float* y, int N, int K, int T, int T_out, int ldt, int P, int dilation, int left,
                           const float* scale, const float* shift, const float* embed, int E, void* stream);
int ps_gated_product_f32(const float* left, const float* right, float* y, int N, int H, int T, int ldt,
                         const ps_prologue* pro_left, const ps_prologue* pro_right, void* stream);

/* apply_tf_masks(real, real) as a kernel of its own (base_nn.py:52-54): out = feats * act(mask) over `rows` rows of ldt
 * frames -- for encoders whose decoder does not fuse the product (the STFT front end with a real mask). */
int ps_real_mask_f32(const float* feats, const float* mask, float* out, int64_t rows, int ldt, int mask_act, void* stream);

/* Magnitude lobe of the speaker branch (lobe/trivial.py:21-59) on the [re rows; im rows] layout [N][2*half][ldt]:
 * y[n][h][t] = sqrt(re^2 + im^2 + 1e-8) of bin h + drop_first (kind 0; kind 1: log1p of it; kind 2: the power re^2 + im^2;
 * kind 3: power + 1e-8 -- ConvMelSpectrogram's "Magnitude" input to the mel projection, lobe/encoder.py:529-536);
 * y is [N][half-drop_first][ldt]. */
int ps_magnitude_f32(const float* x, float* y, int N, int half, int drop_first, int kind, int T, int ldt, void* stream);
/* SpecAugment's masked fill (/root/reference/puresound/nnet/lobe/trivial.py:306-335: the span itself comes from
 * torchaudio.functional.mask_along_axis' two host-side random draws): y = x [N][rows][ld] with rows [lo, hi) of every
 * utterance (axis = 1) or frames [lo, hi) of every row (axis = 2) replaced by `value`; ld % 4 == 0, x == y allowed. */
int ps_fill_span_f32(const float* x, float* y, int N, int rows, int ld, int axis, int lo, int hi, float value, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2-D convolutional maskers (Unet / UnetTcn / DPCRN: unet.py:13-557, dpcrn.py:11-213).  A 4-D activation
 * [N][CH][F][T] is stored as [N][CH*F] rows of ld frames; Conv2d / ConvTranspose2d run as ps_conv1x1_f32 GEMMs
 * (M = Cout, K = Cin*kf*kt, "frames" = (f, t) flattened = Fout*ld) over the unfolded taps.
 *
 * ps_unfold2d_f32: y[n][(ci*kf + jf)*kt + jt][fo*ld + t] = x[n][ci][fi][ti], 0 for ti outside [0, T_in), fi outside
 *   [0, Fin) or t >= T (T_in = frames the input holds, T = frames to produce: T_in + dil_t*(kt-1) for an untrimmed
 *   transposed convolution, whose gLN statistics the reference takes before trimming).
 *   transposed = 0 (nn.ZeroPad2d + nn.Conv2d, unet.py:112-128):  fi = fo*stride_f + jf*dil_f - pad_f,
 *                                                               ti = t + jt*dil_t - pad_t
 *   transposed = 1 (nn.ConvTranspose2d + the time trim, unet.py:139-165,252-256):
 *                  fi = (fo + pad_f - jf*dil_f) / stride_f when divisible,  ti = t + pad_t - jt*dil_t
 *                  (pad_t = the trim shift: transpose_t_size-1 with transpose_delay, else 0)
 *   The input channels are the C1 channels of x1 followed by the C2 channels of x2 (the decoder's
 *   torch.cat([x, skip], 1), unet.py:250; C2 = 0: one tensor).
 * ps_activation_f32: in place on rows of ld frames, kind 0 none, 1 relu, 2 prelu (one shared slope), 3 mish,
 *   4 sigmoid, 5 tanh (lobe/activation.py); pad frames are cleared.
 * ------------------------------------------------------------------------------------------- */
int ps_unfold2d_f32(const float* x1, int C1, const float* x2, int C2, float* y, int N, int Fin, int T_in, int T, int ld,
                    int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f, int pad_t, int Fout, int transposed,
                    void* stream);
/* ps_conv2d_f32: the same convolution WITHOUT materialising the taps (implicit GEMM: the MFMA B operand is gathered from
 * the input rows), bias and activation in the epilogue: y[n][m][fo][t] = act(bias[m] + sum_k W[m][k] * taps[k][fo][t]) with
 * the tap definition, two-source channel concat and T_in / T convention of ps_unfold2d_f32; wt is the packed transposed
 * weight of ps_conv1x1_f32 over K = (C1+C2)*kf*kt (eval BatchNorm2d folded in by the caller); act as ps_activation_f32.
 * Pad frames of y are cleared.  K <= 4096 (PS_E_UNSUPPORTED beyond: use the unfold path). */
int ps_conv2d_f32(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias, float* y, int N,
                  int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f,
                  int pad_t, int Fout, int transposed, int act, const float* slope, void* stream);
/* The same convolution without activation, leaving partial (sum, sum of squares) of its outputs over the valid frames behind:
 * ostats [N][ps_conv2d_stats_parts(M, Fout, ld)][2] doubles -- the statistics of the global LayerNorm (gLN) that follows the
 * convolution in Unet / UnetTcn(norm_type="gLN") (unet.py:100-165, lobe/norm.py), consumed by ps_norm_activation_f32
 * through a PS_NORM_GLOBAL prologue.  One partial per workgroup in its own slot (deterministic). */
int ps_conv2d_stats_parts(int M, int Fout, int ld);
/* ps_conv2d_f32 / ps_conv2d_stats_f32 in the fp16x2 arithmetic (two fp16 terms per operand, three v_mfma_f32_16x16x32_f16
 * products, fp32 accumulation; the activations' range is found by the kernel itself, wave by wave).  wimg: the host-packed
 * image of 2^w_exp W: [channel tiles of MT][ceil(K/32)][2 planes][MT/16][4][16][8] halves with MT = 32 (M <= 32), 64 (M <= 64)
 * or 128: element (tile, chunk, pl, rb, kg, row, e) = plane pl of 2^w_exp W[tile*MT + 16 rb + row][32 chunk + 8 kg + e] (zero
 * outside W; w_exp puts max |2^w_exp W| into [2^13, 2^14)).  ostats: NULL, or the partial statistics of ps_conv2d_stats_f32
 * (act is then applied to the stored values as given; pass 0 in front of a gLN). */
int ps_conv2d_f16x2_f32(const float* x1, int C1, const float* x2, int C2, const void* wimg, int w_exp, const float* bias,
                        float* y, int N, int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f, int dil_f,
                        int dil_t, int pad_f, int pad_t, int Fout, int transposed, int act, const float* slope, double* ostats,
                        void* stream);
int ps_conv2d_stats_f32(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias, float* y, int N,
                        int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f,
                        int pad_t, int Fout, int transposed, double* ostats, void* stream);
int ps_activation_f32(float* x, int kind, const float* slope, int64_t rows, int T, int ld, void* stream);
/* gLN over [CH, F, T] (GlobLN on a 4-D map, lobe/norm.py:20-34) followed by an activation, in place on [N][rows_per_utt]
 * rows of ld frames (row r belongs to channel r / rows_per_channel).  pro = PS_NORM_GLOBAL with the producing GEMM's
 * partial statistics, count = the number of VALID elements per utterance; corr_sum / corr_sq = what the pad columns
 * contributed to those statistics (zero taps there: the GEMM output is its bias, so sum_c bias[c] * F * (ld - T) and the
 * same with bias^2).  pro = PS_NORM_AFFINE (a folded eval BatchNorm: gamma / beta are its per-channel scale / shift)
 * needs no statistics. */
int ps_norm_activation_f32(float* x, const ps_prologue* pro, double corr_sum, double corr_sq, int rows_per_channel,
                           int kind, const float* slope, int N, int rows_per_utt, int T, int ld, void* stream);
/* partial (sum, sum of squares) over the valid frames of each utterance's rows: stats [N][ps_row_stats_parts()][2] fp64,
 * the slab layout the PS_NORM_GLOBAL prologues and ps_norm_activation_f32 read -- GlobLN called on its own
 * (lobe/norm.py:20-34) is ps_row_stats_f64 + ps_norm_activation_f32 */
int ps_row_stats_parts(void);
int ps_row_stats_f64(const float* x, double* stats, int N, int rows, int T, int ld, void* stream);
/* y = a + b over `count` floats (the additive skip connections of Unet(skip_conv=True), unet.py:249); y may alias a or b */
int ps_add_f32(const float* a, const float* b, float* y, int64_t count, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Self-attention of the DPARN bottleneck (MhaSelfAttenLayer / nn.MultiheadAttention(bias=False), lobe/attention.py:38-232;
 * dparn.py:55-80).  qkv [N][3E][ld] = in_proj_weight applied by ps_conv1x1_f32 (q rows | k rows | v rows, head h = rows
 * h*dh..).  A sequence is (n, q), q < Q; its position p sits at frame q*q_stride + p*pos_stride (DPARN: q = frame t,
 * positions = the F frequency rows, pos_stride = ld).  out [N][E][ld] = concat_h softmax(q k^T / sqrt(dh)) v; causal = 1
 * restricts keys to positions <= the query's.  ps_add_position_f32 adds the sinusoidal table pe [L][E]
 * (PositionalEncoding, lobe/attention.py:8-35) to every sequence.
 * ------------------------------------------------------------------------------------------- */
int ps_self_attention_f32(const float* qkv, float* out, int N, int E, int heads, int Q, int q_stride, int L,
                          int pos_stride, int ld, int causal, void* stream);
int ps_add_position_f32(const float* x, const float* pe, float* y, int N, int E, int Q, int q_stride, int L,
                        int pos_stride, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Recurrent maskers (DPRNN dprnn.py:111-191, SkiM skim.py:198-229,45-114,410-469, StreamingSkiM
 * streaming/skim_inference.py:41-252).  They run on the same channel-major padded layout, so their Linear layers
 * (LSTM input projections, proj, FiLM convs, output_fc) are ps_conv1x1_f32 calls; what is left is below.
 *
 * ps_lstm_f32: the recurrence of nn.LSTM(num_layers=1, batch_first=True), one or two directions.
 *   gx    [N][D*4H][ldt]  W_ih x + b_ih + b_hh for every frame (rows: direction, then gates i,f,g,o, then unit)
 *   whh_t [D][H][4H]      weight_hh_l0(_reverse) transposed
 *   A sequence is (n, q), q < Q; its step s reads/writes frame q*q_stride + s*step_stride (direction 1 walks the
 *   steps backwards):  intra-segment pass Q=S, q_stride=K, steps=K, step_stride=1;  inter-segment pass Q=K,
 *   q_stride=1, steps=S, step_stride=K;  streaming step Q=streams, q_stride=1, steps=1.
 *   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c');  hout [N][D*H][ldt] gets h' at the step's frame.
 *   States use the "state layout" [N][D*H][ldq], one frame per sequence: h0/c0 initial (NULL = zeros), h_last /
 *   c_last final (NULL = not wanted; may alias h0/c0 when state_shift == 0).  state_shift = 1 starts flat sequence
 *   b = n*Q+q from the state stored for b-1 and sequence 0 from zeros -- MemLSTM's causal hand-over including its
 *   cross-utterance leak (skim.py:102-109).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* gx;
  const float* whh_t;
  const float* h0;
  const float* c0;
  float* hout;
  float* h_last;
  float* c_last;
  int N, H, D, Q, q_stride, steps, step_stride, ldt, ldq, state_shift;
} ps_lstm_args;
int ps_lstm_f32(const ps_lstm_args* args, void* stream);
/* The recurrences of nn.RNN (tanh) and nn.GRU (num_layers = 1) -- SingleRNN(rnn_type = "RNN" | "GRU"),
 * /root/reference/puresound/nnet/lobe/rnn.py:19-35; no recipe builds them, the kernel is the plain one.  ps_lstm_args with
 * G = H (RNN) or 3H (GRU: gates r, z, n) gate rows per direction in gx [N][D*G][ldt] and whh_t [D][H][G]; c0 / c_last must be
 * NULL, state_shift 0.  gx = W_ih x + b_ih + b_hh, except the GRU's n gate whose hidden bias stays inside the reset product:
 * bhn [D][H] (n = tanh(gx_n + r * (W_hn h + b_hn)), h' = (1 - z) n + z h). */
#define PS_RNN_TANH 0
#define PS_RNN_GRU 2
int ps_rnn_f32(const ps_lstm_args* args, int kind, const float* bhn, void* stream);
/* The same recurrence with the product W_hh h in the fp16x2 arithmetic of ps_conv1x1_f16x2_f32 (two fp16 terms per
 * operand, three MFMA products, fp32 accumulation on top of the fp32 pre-activations; error <= 2^-21 of sum |W||h| per
 * gate and step) where a kernel for it exists -- H = 64 with 20 consecutive 16-byte-aligned steps per sequence (the
 * intra-segment pass of DPRNN(seg_size=20), dprnn.py:154-160) or with strided steps and fewer than 4096 sequences (its
 * inter-segment pass, dprnn.py:162-171) -- and exactly ps_lstm_f32 for every other shape.  h0 is taken as an LSTM
 * output (|h0| <= 1; the two-term form holds up to |h0| < 32). */
int ps_lstm_f16x2_f32(const ps_lstm_args* args, void* stream);
/* The recurrence of ps_lstm_f16x2_f32 for H = 128 with gx FRAME-MAJOR, [N][ldt][ldm >= D*4H] (ps_conv1x1_f16x2_fmajor_f32): 16
 * sequences per workgroup, v_mfma_f32_16x16x32_f16, the pre-activations streamed HBM -> LDS by DMA four steps ahead.
 * The bottleneck LSTMs of DPCRN (dpcrn.py:34-81: bidirectional along frequency, Q = T, q_stride = 1, step_stride = ld;
 * unidirectional along time, Q = F, q_stride = ld, step_stride = 1).  hout as ps_lstm_f32 ([N][D*H][ldt]).  No initial or
 * final states (h0, c0, h_last, c_last must be NULL).  step_stride = 1 needs 16-byte-aligned rows and q_stride and, when
 * steps % 4 != 0, D = 1 and room for the padded group inside the row (those frames are written as zeros).
 * ps_lstm_fmajor_ok(args, ldm) = 1 when a launch qualifies, else PS_E_UNSUPPORTED. */
int ps_lstm_fmajor_ok(const ps_lstm_args* args, int ldm);
/* The same for H = 256 (SkiM's segment LSTMs, /root/reference/puresound/nnet/skim.py:45-114) and H = 192 (the speaker LSTM of
 * tse_skim_v1): W_hh as two fp16 terms is 1 MiB per direction and fits no CU, so it is streamed from L2 every step in
 * MFMA-fragment order.  whh_image: the host-packed image [D][H/32][H/32][2][4][2][64][8] halves (wave, k-step, row block, gate,
 * plane, lane, k): element (w, ks, rb, g, pl, lane, e) =
 * plane pl of 2^(13 - e_d) * W_hh[g*H + 32 w + 16 rb + lane % 16][32 ks + 8 (lane / 16) + e], e_d from frexp(max |W_hh| of
 * direction d); acc_scale [D] (host memory) = 2^(13 - e_d) * 1024.  args->whh_t is not read.  Initial / final states and
 * state_shift as ps_lstm_f32.  hout [N][D*H][ldt]. */
int ps_lstm_fmajor_h256_ok(const ps_lstm_args* args, int ldm);
int ps_lstm_fmajor_h256_f16x2_f32(const ps_lstm_args* args, int ldm, const void* whh_image, const float* acc_scale, void* stream);
/* The same recurrence for FEW sequences and many steps (the speaker LSTM of tse_skim_v1: SingleRNN over all frames,
 * lobe/rnn.py:9-55; SkiM's Mem-LSTMs, skim.py:117-170): every group of 16 sequences is spread over H / 32 workgroups on
 * CUs of their own, each holding its 32 hidden units' rows of W_hh in registers for the whole launch and exchanging h' with
 * the others through L2 every step (one counter per group, bounded polls: a launch whose workgroups are not all resident at
 * once -- the launcher refuses such grids -- would raise the error word behind the counters instead of hanging).  Same
 * arguments, weight image and results class as ps_lstm_fmajor_h256_f16x2_f32, plus a scratch buffer of
 * ps_lstm_fmajor_coop_workspace_bytes(args, ldm) bytes, 256-byte aligned; 0 bytes = the launch does not qualify (more than
 * CUs / (D * H / 32) groups, or fewer than two steps): PS_E_UNSUPPORTED, take the streamed kernel.
 * ONE such launch in flight per device at a time: its workgroups wait for each other, so two of them on different streams can
 * each hold CUs the other needs until both give up (NaN results and the error word, not a hang).  The Python mirror's batch
 * lanes (hip_streams > 1) therefore take the streamed kernel. */
size_t ps_lstm_fmajor_coop_workspace_bytes(const ps_lstm_args* args, int ldm);
int ps_lstm_fmajor_coop_f16x2_f32(const ps_lstm_args* args, int ldm, const void* whh_image, const float* acc_scale,
                                  void* workspace, size_t workspace_bytes, void* stream);
int ps_lstm_fmajor_f16x2_f32(const ps_lstm_args* args, int ldm, void* stream);

/* 50 % overlapped segmentation of the dual-path maskers (SplitMerge.split / merge, lobe/trivial.py:178-241; SkiM.split /
 * merge, skim.py:334-408) on rows of frames:
 *   merge = 0: dst frame s*K + k = src frame (s/2)*K + k + (s odd ? K/2 : 0) - K/2, zero outside [0, T_src); T_dst =
 *              S*K with S even = 2 * (T_src + rest + K/2) / K, rest = K - (K/2 + T_src % K) % K
 *   merge = 1: dst frame t = (even-segment cover + odd-segment cover) / 2, T_dst = the original frame count. */
int ps_segment_overlap_f32(const float* src, float* dst, int64_t rows, int T_src, int ld_src, int T_dst, int ld_dst,
                           int K, int merge, void* stream);

/* Fused forms of the streaming frame step (StreamingSkiM.step_frame, streaming/skim_inference.py:176-218; frames =
 * concurrent streams).  All three share the short-row GEMM (16 output channels x <= 64 frames per workgroup).
 *
 * ps_film_conv_f32: FiLM conditioning after its input norm (lobe/trivial.py:160-167):
 *     y[c] = (Ws x + rs)[c] * x[c] + (Wb x + rb)[c]
 *   wt_pairs = packed weight of the [2C][C] matrix whose rows (2c, 2c+1) are (cond_scale row c, cond_bias row c)
 *   restricted to the feature columns; res_pairs [N][2C][ldt] = the embedding columns' per-frame contribution in the
 *   same row order (NULL = none).
 * ps_lstm_gates_cell_f32: gates = W [x; h] + b, then the LSTM cell, for a one-frame SegLSTM step
 *   (skim.py:215-222 with seg_size 1).  xh [N][K][ldt] holds x on top of h_{t-1} (K = C + H); wt_units / bias_units
 *   are [W_ih | W_hh] and b_ih + b_hh with rows reordered unit-major (row 4u+g = gate g of unit u, gates i,f,g,o).
 *   c [N][H][ld_state] is updated in place, h' goes to h [N][H][ld_state], which must NOT alias the h rows of xh.
 * ps_proj_layernorm_f32: y = res + LN(W x + b; gamma, beta, eps) (skim.py:223-227), optionally also
 *   y2 = LN(y; gamma2, beta2, eps2) (the next block's FiLM input norm) and x_copy = x (hands h' back to the xh block).
 *   res_inside = 1 moves the residual inside the norm: y = LN(W x + b + res) (the post-norm transformer layer of
 *   lobe/attention.py:211-226).  M <= 256. */
int ps_film_conv_f32(const float* x, const float* wt_pairs, const float* res_pairs, float* y, int N, int C, int T,
                     int ldt, void* stream);
int ps_lstm_gates_cell_f32(const float* xh, const float* wt_units, const float* bias_units, float* c, float* h, int N,
                           int K, int H, int T, int ldt, int ld_state, void* stream);
int ps_proj_layernorm_f32(const float* x, const float* wt, const float* bias, const float* gamma, const float* beta,
                          float eps, const float* res, float* y, const float* gamma2, const float* beta2, float eps2,
                          float* y2, float* x_copy, int res_inside, int N, int K, int M, int T, int ldt, void* stream);
/* The same three operators for SEVERAL independent one-utterance problems of one shape in ONE launch each (blockIdx.y =
 * the problem): the streaming harness runs the (hop, block) cells of a chunk as a wavefront (cell (h, i) needs (h, i-1) and
 * (h-1, i) only), and the cells of an anti-diagonal differ in nothing but their pointers -- block i's weights and state, the
 * hop's rows.  `cells` is a HOST array of ncells <= PS_MAX_CELLS entries, copied into the kernel arguments; fields as the
 * arguments of the same name above with N = 1; results are bit-identical to ncells separate calls. */
#define PS_MAX_CELLS 8
typedef struct ps_film_cell {
  const float* x;
  const float* wt_pairs;
  const float* res_pairs; /* may be NULL */
  float* y;
} ps_film_cell;
typedef struct ps_gates_cell {
  const float* xh;
  const float* wt_units;
  const float* bias_units; /* may be NULL */
  float* c;
  float* h;
} ps_gates_cell;
typedef struct ps_projln_cell {
  const float* x;
  const float* wt;
  const float* bias; /* may be NULL */
  const float* gamma;
  const float* beta;
  const float* res; /* may be NULL */
  float* y;
  const float* gamma2; /* with y2 */
  const float* beta2;
  float* y2;     /* may be NULL */
  float* x_copy; /* may be NULL */
  float eps, eps2;
} ps_projln_cell;
int ps_film_conv_cells_f32(const ps_film_cell* cells, int ncells, int C, int T, int ldt, void* stream);
int ps_lstm_gates_cell_cells_f32(const ps_gates_cell* cells, int ncells, int K, int H, int T, int ldt, int ld_state,
                                 void* stream);
int ps_proj_layernorm_cells_f32(const ps_projln_cell* cells, int ncells, int res_inside, int K, int M, int T, int ldt,
                                void* stream);
/* The same on long rows with the partial maxima of |y| as a by-product: y_amax [N][ps_proj_layernorm_amax_parts(T)] is what
 * a following ps_conv1x1_f16x2_f32 takes as x_amax (the LSTM input projections of DPRNN / SkiM in the fp16x2 arithmetic).
 * Only the row kernel produces them (T >= 128, y2 == x_copy == NULL); otherwise PS_E_UNSUPPORTED. */
int ps_proj_layernorm_amax_parts(int T);
int ps_proj_layernorm_amax_f32(const float* x, const float* wt, const float* bias, const float* gamma, const float* beta,
                               float eps, const float* res, float* y, const float* gamma2, const float* beta2, float eps2,
                               float* y2, float* x_copy, int res_inside, int N, int K, int M, int T, int ldt, float* y_amax,
                               void* stream);

/* Streaming harness overlap-add (egs/tse/demo/utils.py:121-128): out[b][j] = (tail[b][j] + cur[b][j]) / 2 for
 * j < overlap (tail = the last `overlap` samples of the running output, row stride ld_tail), cur[b][j] otherwise. */
int ps_overlap_average_f32(const float* tail, int ld_tail, const float* cur, float* out, int B, int win, int overlap,
                           void* stream);

/* The same harness for ALL hops of a chunk in two launches (replaces the hop-by-hop window shift and overlap-add of
 * DemoTseNet.streaming_inference_chunk, egs/tse/demo/utils.py:100-128; win = 2 * hop):
 *   ps_stream_windows_f32  wins[i][b * win + j] = sig_b[i * hop + j], sig_b = queue[b][hop .. win) ++ chunk[b][0 .. hops * hop)
 *                          (queue [B][win] = the previous window, chunk [B][hops * hop]) -- one "utterance" per hop for the
 *                          encoder;
 *   ps_stream_overlap_f32  frames [hops][B][win] (the decoder's output) -> blocks[b][i * hop + j] = (prev + frames[i][b][j]) / 2
 *                          with prev = tail[b][j] (i = 0) or frames[i-1][b][hop + j]; then tail[b][:] = frames[hops-1][b][hop:]
 *                          and queue[b][:] = wins[hops-1][b] (the last window).  tail [B][hop], blocks [B][hops * hop]. */
int ps_stream_windows_f32(const float* queue, const float* chunk, float* wins, int B, int hops, int win, int hop, void* stream);
int ps_stream_overlap_f32(const float* frames, const float* wins, float* tail, float* blocks, float* queue, int B, int hops,
                          int win, int hop, void* stream);

/* One cell update per (unit, frame) from COMPLETE gate pre-activations gates [N][D*4H][ld_gates] (W_ih x + W_hh h + both
 * biases: the streaming step puts [x; h] on the K axis of one ps_conv1x1_f32):  c' = sig(f) c + sig(i) tanh(g) in
 * place in c [N][D*H][ld_state], h' = sig(o) tanh(c') into h [N][D*H][ld_state].  (SegLSTM with a one-frame
 * sequence, streaming/skim_inference.py:198-207.) */
int ps_lstm_cell_f32(const float* gates, float* c, float* h, int N, int H, int D, int T, int ld_gates, int ld_state,
                     void* stream);

/* y = [res +] [mul *] act( LN_C(x) * gamma + beta ): statistics over the C channels of each frame, biased two-pass
 * variance, 1/sqrt(var + eps).  nn.LayerNorm(C) on [.., C] rows (eps 1e-5; dprnn.py:157,171, skim.py:85-98,226,
 * FiLM's input norm lobe/trivial.py:160) and ChanLN (eps 1e-8, lobe/norm.py:37-50).  act: PReLU when prelu_slope
 * is given (one shared slope), then sigmoid when `sigmoid` (Gate, lobe/trivial.py:75-104). */
int ps_chan_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, const float* prelu_slope,
                          int sigmoid, const float* mul, const float* res, float* y, int N, int C, int T, int ldt,
                          void* stream);

/* FiLM (lobe/trivial.py:162-167): y[n][c][t] = sb[n][c][t] * x[n][c][t] + sb[n][C + c][t]; sb [N][2C][ldt] is the
 * output of cond_scale / cond_bias stacked into one 1x1 conv. */
int ps_film_apply_f32(const float* x, const float* scale_bias, float* y, int N, int C, int T, int ldt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One "normal" TCN block and the whole Conv-TasNet masker (conv_tasnet.py:67-90, 338-359).
 * All weights are device pointers in the packed forms documented above.
 * ------------------------------------------------------------------------------------------- */
typedef struct ps_tcn_block {
  int C, H, P, dilation, causal;
  int in_norm, dw_norm, pw_norm; /* PS_NORM_GLOBAL or PS_NORM_AFFINE */
  const float* in_wt;            /* kernel layout of in_conv.0.weight[:, :C] (see ps_conv1x1_f32) */
  const float* in_embed_w;       /* [H][E] or NULL */
  int E;
  const float *in_gamma, *in_beta, *in_slope;
  const float *dw_w, *dw_b, *dw_gamma, *dw_beta, *dw_slope;
  const float *pw_wt, *pw_b, *pw_gamma, *pw_beta, *pw_slope; /* pw_wt: kernel layout */
  const float *out_wt, *out_b;                                /* out_wt: kernel layout */
  /* matrix-pipe arithmetic of the three 1x1 convs: 0 = exact fp32 MFMA (the *_wt pointers above);
   * 1 / 3 = ps_conv1x1_bf16_f32 with that many planes over the plane-packed weights below; 2 = fp16x2 (see the end) */
  int gemm_planes;
  const void *in_wb, *pw_wb, *out_wb;
  /* with gemm_planes = 1 (BASELINE's "bf16" configurations): keep the block's three hidden maps -- workspace only,
   * never visible to the caller -- as bf16 rows: written by the GEMM epilogue / the depthwise kernel, read by the next
   * prologue; statistics, bias, accumulation and the block's input / output (the residual stream) stay fp32 */
  int hidden_bf16;
  /* gemm_planes = 2 ("fp16x2", ps_conv1x1_f16x2_f32): in_wb / pw_wb / out_wb hold the two-plane fp16 images of
   * 2^w_exp[i] * W (i = 0 / 1 / 2: in, pointwise, out); dw_* / pw_* = max |gamma|, max |beta| of the norms in front of
   * the pointwise and output convs (both PS_NORM_GLOBAL: the bound on the normalised values gives the activation scale);
   * the residual stream's range travels from out_conv to the next in_conv as partial maxima inside the workspace */
  int w_exp[3];
  float dw_gmax, dw_bmax, pw_gmax, pw_bmax;
  /* gemm_planes = 1 with hidden_bf16, inside ps_conv_tasnet_bf16_rows: optional two-plane fp16 images of the three
   * weights (as for gemm_planes = 2, with w_exp / dw_* / pw_* above filled in the same way).  Launches large enough for
   * the register-B kernel then run ps_conv1x1_f16_rows (bf16 rows, one fp16 product per multiply-add) instead of the
   * planes = 1 kernels; NULL keeps the latter. */
  const void *in_wf, *pw_wf, *out_wf;
} ps_tcn_block;

/* ps_conv1x1_bf16_f32 / ps_dwconv_f32 with bf16 activation rows (x_bf16 / y_bf16 != 0: the buffer holds
 * [N][channels][ldt] bf16 instead of fp32; planes must be 1; with y_bf16 a residual is bf16 rows as well; the depthwise
 * form is built for P = 3).  Used inside ps_conv_tasnet_f32 for the hidden maps of a block with hidden_bf16. */
int ps_conv1x1_bf16_io(const void* x, int x_bf16, const void* wt_planes, void* y, int y_bf16, int N, int K, int M, int T,
                       int ldt, int planes, const ps_prologue* pro, const float* bias, const float* bias_n,
                       const float* res, double* ostats, void* stream);
int ps_dwconv_io(const void* x, int x_bf16, const float* w, const float* b, void* y, int y_bf16, int N, int H, int T,
                 int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats, void* stream);
/* The depthwise convolution leaving the partial maxima of |y| over the valid frames behind instead of the statistics:
 * y_amax [N][ps_dwconv_stats_parts(H, T)] -- the input range of a following ps_conv1x1_f16x2_f32 whose prologue is a
 * per-channel affine map (an eval BatchNorm) rather than a global norm (ps_f16x2_range.amax_mul / amax_add).  fp32 rows,
 * P = 3 and 2 * dilation <= 256 (ps_dwconv_amax_ok = 1); otherwise PS_E_UNSUPPORTED: ps_dwconv_f32, then ps_absmax_f32. */
int ps_dwconv_amax_ok(int P, int dilation, int left);
int ps_dwconv_amax_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int T, int ldt, int P,
                       int dilation, int left, const ps_prologue* pro, float* y_amax, void* stream);


/* ps_conv1x1_f32's contract in the "fp16x2" arithmetic: every operand as two fp16 terms (22 significant bits), three
 * products W0x0 + W0x1 + W1x0 on v_mfma_f32_32x32x16_f16, fp32 accumulation: half the matrix-pipe work of the
 * three-plane bf16 split at <= 3 * 2^-22 relative error per product.  fp16 has five exponent bits, so both operands
 * are brought into its range by powers of two (undone exactly in the epilogue):
 *   weights      wt_planes = the two-plane image of 2^w_exp * W ([ceil(M/256)][ceil(K/16)][2][256][16] fp16,
 *                ps_conv1x1_bf16_weight_bytes(M, K, 2) bytes); the packer picks w_exp so that max |2^w_exp W| lies in
 *                [2^13, 2^14);
 *   activations  x_bound > 0: a host-known bound on |f(x)| (behind a global norm: max|gamma| sqrt(count) + max|beta|, times
 *                max(1, |slope|) when a PReLU follows: f includes it),
 *                the kernel scales by the power of two that puts it below 2^15;
 *                else x_amax != NULL: [N][x_amax_parts] partial maxima of |x| left by the producer (y_amax of the
 *                launch before, or ps_absmax_f32): per utterance, max |x| goes to [2^14, 2^15);
 *                else 2^-2 behind a normalising prologue and 2^-4 otherwise (|values| beyond 2.6e5 / 1e6 overflow to
 *                inf / NaN in y; small inputs lose relative precision: absolute resolution 2^-23 / 2^-21).
 *                With amax_mul > 0 the maxima are those of the producer's output BEFORE the prologue f (a per-channel
 *                affine map -- an eval BatchNorm -- and a PReLU): the range of f(x) is taken as amax_mul * max|x| + amax_add
 *                (max |scale| and max |shift| of the map, times max(1, |slope|)); 0 = the maxima as they are.
 *   y_amax       optional, [N][ps_conv1x1_stats_parts(M, T)]: partial maxima of |y| over the valid frames. */
typedef struct ps_f16x2_range {
  int w_exp;
  float x_bound;
  const float* x_amax;
  int x_amax_parts;
  float* y_amax;
  float amax_mul, amax_add;
} ps_f16x2_range;
int ps_conv1x1_f16x2_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N, int K, int M,
                         int T, int ldt, const ps_prologue* pro, const float* bias, const float* bias_n,
                         const float* res, double* ostats, void* stream);

/* The 1x1 convolution with every activation row (x, y, the residual) stored as bf16 and ONE fp16 product per
 * multiply-add: x is rounded to fp16 behind its prologue (a bf16-stored value has 8 significant bits, fp16 keeps 11),
 * the weights enter as the first plane of their fp16x2 image (11 bits against bf16's 8), accumulation, bias, statistics
 * and the range handling are those of ps_conv1x1_f16x2_f32 (rng as there; a range is mandatory).  Runs on the
 * register-B kernel only: ps_conv1x1_f16_rows_ok(N, K, M, T) says whether a launch qualifies (K % 32 == 0,
 * M % 256 == 0, enough tiles for the chip); otherwise PS_E_UNSUPPORTED and the caller takes ps_conv1x1_bf16_io.
 * Replaces the same reference op as ps_conv1x1_f32 (/root/reference/puresound/nnet/conv_tasnet.py:43-90) in the
 * "bf16 storage / fp32 accumulate" arithmetic BASELINE.json names for config 3. */
int ps_conv1x1_f16_rows_ok(int N, int K, int M, int T);
int ps_conv1x1_f16_rows(const void* x, const void* wt_planes, const ps_f16x2_range* rng, void* y, int N, int K, int M,
                        int T, int ldt, const ps_prologue* pro, const float* bias, const float* bias_n, const void* res,
                        double* ostats, void* stream);
/* ps_conv1x1_f16x2_f32 (no prologue, residual or statistics) writing its output FRAME-MAJOR: y [N][ldt][ldm], ldm >= M a
 * multiple of 4 -- the M outputs of a frame are consecutive -- instead of [N][M][ldt] (ldm = M is what the host side uses; a
 * skew between frames made no measurable difference).  This is the layout ps_lstm_fmajor_f16x2_f32 reads its
 * gate pre-activations in: the LSTM input projection W_ih x + b of nn.LSTM (/root/reference/puresound/nnet/lobe/rnn.py:
 * 9-55 as used by dpcrn.py:34-81) written so that one step of 16 sequences is 16 contiguous 2 KiB runs.  Register-B kernel
 * only: ps_conv1x1_f16x2_fmajor_ok says whether a launch qualifies (as ps_conv1x1_f16_rows_ok, and ldm*ldt*4 < 2^31). */
int ps_conv1x1_f16x2_fmajor_ok(int N, int K, int M, int T, int ldt, int ldm);
int ps_conv1x1_f16x2_fmajor_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N, int K,
                                int M, int T, int ldt, int ldm, const float* bias, void* stream);
/* Projection + LayerNorm over channels + skip in one launch, C = 128 output channels:
 *   res_inside = 0:  y[n][c][t] = res[n][c][t] + LN_c(W f(x) + b)[c]   (nn.Linear + nn.LayerNorm + the residual behind each
 *                    recurrence of DPRNNblock2D / DPARNblock2D, /root/reference/puresound/nnet/dpcrn.py:56-80, dparn.py:81-108)
 *   res_inside = 1:  y = LN_c(W f(x) + b + res)                        (the post-norm blocks of MhaSelfAttenLayer,
 *                    lobe/attention.py:202-232: out_proj + residual + norm1, feed-forward + residual + norm2)
 * The GEMM runs in the fp16x2 arithmetic of ps_conv1x1_f16x2_f32 with the LayerNorm (two-pass variance, eps inside the root,
 * as nn.LayerNorm) as its epilogue; f = the prologue `pro` (NULL, a per-channel affine map and / or a PReLU: the ReLU in
 * front of the second feed-forward layer is the PReLU of slope 0).  wt_planes: the fp16x2 image of W zero padded to 256 rows;
 * bias [128] or NULL; gamma, beta [128], 16-byte aligned; res [N][128][ldt] or NULL; rng->y_amax as ps_conv1x1_f16x2_f32
 * with ps_conv1x1_stats_parts(256, T) parts.  Register-B kernel only: ps_conv1x1_f16x2_ln_ok(N, K, C, T) says whether a
 * launch qualifies. */
int ps_conv1x1_f16x2_ln_ok(int N, int K, int C, int T);
int ps_conv1x1_f16x2_ln_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N, int K, int C,
                            int T, int ldt, const ps_prologue* pro, const float* bias, const float* gamma, const float* beta,
                            float eps, const float* res, int res_inside, void* stream);
/* partial maxima of |x| over the valid frames: amax [N][ps_absmax_parts()] */
int ps_absmax_parts(void);
int ps_absmax_f32(const float* x, float* amax, int N, int C, int T, int ldt, void* stream);

/* bytes of scratch ps_conv_tasnet_f32 needs for a batch (3 hidden maps + stats + embed bias) */
size_t ps_conv_tasnet_workspace_bytes(int N, int C, int H, int T);

/* x_in [N][C][ldt] is read only; x_out [N][C][ldt] receives the mask logits (pre-constraint).
 * dvec [N][E] may be NULL; blocks_host is a HOST array (it only carries launch arguments). */
int ps_conv_tasnet_f32(const ps_tcn_block* blocks_host, int n_blocks, const float* x_in, float* x_out,
                       const float* dvec, int embed_norm, int N, int T, int ldt, void* workspace,
                       size_t workspace_bytes, void* stream);
/* the same with the range of x_in handed over for blocks in the fp16x2 arithmetic: x_amax [N][x_amax_parts] holds, per
 * utterance, values whose maximum is >= max |x_in[n]| (exact partial maxima, or any upper bound -- e.g. the encoder's
 * max |wav| * max row sum of |w|); NULL = ps_conv_tasnet_f32 (one ps_absmax_f32 pass over x_in when block 0 needs it) */
int ps_conv_tasnet_ranged_f32(const ps_tcn_block* blocks, int n_blocks, const float* x_in, float* x_out,
                              const float* dvec, int embed_norm, int N, int T, int ldt, void* workspace,
                              size_t workspace_bytes, const float* x_amax, int x_amax_parts, void* stream);

/* The same masker with the residual stream stored as bf16 rows too: x_in / x_out are [N][C][ldt] bf16.  Every block must
 * be in the bf16 arithmetic (gemm_planes = 1, hidden_bf16 = 1): bf16 products and rows, fp32 accumulation, statistics,
 * bias and norm parameters -- what BASELINE.json names for its config 3 ("bf16 storage / fp32 accumulate";
 * /root/reference/egs/tse/model.py:95-140 builds the model).  Not an fp32-class result: acceptance l2-rel <= 3e-2. */
int ps_conv_tasnet_bf16_rows(const ps_tcn_block* blocks, int n_blocks, const void* x_in, void* x_out, const float* dvec,
                             int embed_norm, int N, int T, int ldt, void* workspace, size_t workspace_bytes, void* stream);

/* Five moments of an (estimate, reference) waveform pair per row -- sum a, sum b, sum a^2, sum b^2, sum ab over L
 * samples, fp64 -- as per-workgroup partials [N][ps_wave_moments_chunks(L)][5] (the caller adds them up).  Every
 * SDR / SI-SNR variant of the reference's SDRLoss.forward (puresound/nnet/loss/sdr.py:104-183), si_snr (:263-299) and
 * inactive_sdr_loss (:302-322) is algebra on these moments, so the mean / subtract / project / square / sum passes of
 * the reference become one streaming pass.  lda / ldb: row strides in floats. */
int ps_wave_moments_chunks(int L);
int ps_wave_moments_f64(const float* a, const float* b, double* partials, int N, int L, int lda, int ldb, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PURESOUND_HIP_H */
