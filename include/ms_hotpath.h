/*
 * ms_hotpath.h -- C ABI of the MI355X-native myrtlespeech hot path.
 *
 * libms_hotpath.so is the drop-in boundary: plain pointers, sizes and a HIP
 * stream, no torch types.  The reference (MyrtleSoftware/myrtlespeech) has no
 * FFI of its own -- every FLOP on its hot path is a stock PyTorch op called
 * from a torch.nn.Module -- so each entry point below cites the reference
 * call site (file:line under src/myrtlespeech/) whose arithmetic it replaces.
 * The host-side mirror of those modules lives in myrtlespeech_amd/ (Python,
 * as the reference is Python); INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - tensors are dense, row-major, float32 unless stated; lengths are int32;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue
 *     work, they never synchronise (exceptions: ms_rnn_status, ms_prof_read and the greedy ms_rnnt_decode);
 *   - return value: MS_OK or an MS_ERR_* code; ms_last_error() describes the
 *     last failure on the calling thread;
 *   - outputs are caller-allocated; nothing is freed or retained.
 *
 * Operand precision (environment variable MS_PRECISION, read once per process
 * at the first launch; it selects kernels inside ms_rnn_pack / ms_rnn_layer_forward,
 * ms_linear_split_forward and ms_maskconv_cl_*; packed weights are only valid for
 * the mode they were packed in)
 *   unset / "bf16x3"  every float32 operand of the large contractions is split into
 *                     bf16 hi + lo and multiplied as hi*hi + lo*hi + hi*lo with float32
 *                     accumulation (relative error ~2^-17 per product; full-size DS2
 *                     logits within 2.5e-7 of the reference);
 *   "f32"             float32 MFMA everywhere (3.9e-8); the two-stream LSTM hands h to the
 *                     other workgroups with its mantissa LSB used as an epoch tag;
 *   "fp16"            one fp16 pass (1.1e-5 on that network; relative, not a parity mode).
 * Convolutions with few input channels, small GEMMs, CTC loss / gradient and the
 * decoders are float32 in every mode.
 */
#ifndef MS_HOTPATH_H
#define MS_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): MS_PRECISION's default is "f16x3" (fp16 hi + lo operand planes, three fp16 MFMAs; "bf16x3" selects the bf16
 * pairs of versions 1-3): packed weights made by an older library are in another plane format; ms_maskconv_cl_packed_bytes /
 * ms_maskconv_fwin_packed_bytes grew by 256 bytes (the weights' power-of-two scale word); MS_RNN_TIMING_SKIP_PROJECTION,
 * ms_rnn_stack_overlap_ok and ms_rnn_stack_forward are new; ms_rnn_workspace_bytes grew (four projection buffers for K-halves layers).
 * 3 (round 5): ms_ctc_status, ms_rnn_padded_hidden and ms_rnn_hx_preinit (+ MS_RNN_HX_PREINIT, exchange regions) are new; the CTC workspaces start with a 256-byte status region;
 * ms_linear_splitk_workspace_bytes / ms_linear_splitk_forward take `flags` (MS_LINEAR_FEW_ROWS): the K-slice count
 * no longer depends on M.
 * 2 (round 4): ms_prof_read writes MS_PROF_KINDS = 9 entries (was 4 in version 1); the `zero_infinity` argument of the CTC
 * entry points is a bit field {1 = zero_infinity, MS_CTC_LOG_PROBS_IN} and values above 3 are rejected;
 * ms_ctc_loss_backward takes the same bit field; ms_log_softmax_axis_backward is new.  The Python binding refuses a
 * library whose ms_abi_version() differs (myrtlespeech_amd/_lib.py). */
#define MS_ABI_VERSION 4

enum {
  MS_OK = 0,
  MS_ERR_INVALID = 1,     /* bad argument (shape, NULL pointer, unsupported value) */
  MS_ERR_HIP = 2,         /* a HIP runtime call failed */
  MS_ERR_WORKSPACE = 3,   /* workspace too small */
  MS_ERR_TIMEOUT = 4,     /* a persistent kernel gave up waiting for its peers */
  MS_ERR_UNSUPPORTED = 5
};

/* recurrent cell kinds: model/rnn.py:9-12 RNNType (+ model/hard_lstm.py) */
enum { MS_CELL_LSTM = 0, MS_CELL_GRU = 1, MS_CELL_RNN_TANH = 2, MS_CELL_HARD_LSTM = 3 };

/* activation fused into an epilogue: builders/activation.py:31-42 */
enum { MS_ACT_NONE = 0, MS_ACT_CLAMP = 1 /* Hardtanh(lo,hi); ReLU = clamp(0,+inf) */ };

int ms_abi_version(void);
const char* ms_last_error(void);

/* ---- model/cnn.py ------------------------------------------------------- */

/* MaskConv{1,2}d._mask_ (cnn.py:280-293, 425-443): x[n, :, t] = 0 for t >= lens[n],
 * in place.  x is [N, inner, T]. */
int ms_mask_time_(float* x, const int32_t* lens, int N, int inner, int T, void* stream);

/* Bytes of the packed filter bank ms_maskconv_pack writes (Cout = all groups). */
size_t ms_maskconv_packed_bytes(int Cout, int Cin_g, int KF, int KT, int groups);

/* Re-lays torch's Conv weight [Cout, Cin_g, KF, KT] (cnn.py:238-246, 377-385)
 * into the [group][cout-tile][cin*KF+kf][kt (padded even)][32] bank the kernel stages. */
int ms_maskconv_pack(const float* w, void* packed, int Cout, int Cin_g, int KF, int KT, int groups, void* stream);

/* MaskConv2d.forward (cnn.py:445-483) = mask + F.pad + Conv2d (+ the following
 * SeqLenWrapper activation, seq_len_wrapper.py:28-32), fused: frames t >= lens[n]
 * of the input read as 0, SAME padding is index arithmetic (pad_f_l / pad_t_l are
 * the LEFT pads of cnn.py:148-163), bias and clamp in the epilogue.
 * x [N, Cin, Fin, Tin] -> y [N, Cout, Fout, Tout].  MaskConv1d (cnn.py:295-333)
 * is the Fin = KF = SF = 1 case.  lens may be NULL (no masking).  groups >= 1. */
int ms_maskconv_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias, float* y,
                        int N, int Cin, int Fin, int Tin, int Cout, int Fout, int Tout, int KF, int KT, int SF,
                        int ST, int DF, int DT, int pad_f_l, int pad_t_l, int groups, int act, float act_lo,
                        float act_hi, void* stream);

/* Same contract as ms_maskconv_forward for groups == 1 and Cin % 16 == 0 (e.g. the second DS2
 * convolution), computed as a split-bf16 implicit GEMM over channels (x.w ~= x_hi.w_hi +
 * x_lo.w_hi + x_hi.w_lo, f32 accumulate).  The workspace receives the channels-last bf16
 * planes of the input. */
size_t ms_maskconv_cl_packed_bytes(int Cout, int Cin, int KF, int KT);
int ms_maskconv_cl_pack(const float* w, void* packed, int Cout, int Cin, int KF, int KT, void* stream);
size_t ms_maskconv_cl_workspace_bytes(int N, int Cin, int Fin, int Tin);
int ms_maskconv_cl_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias, float* y,
                           int N, int Cin, int Fin, int Tin, int Cout, int Fout, int Tout, int KF, int KT, int SF,
                           int ST, int DF, int DT, int pad_f_l, int pad_t_l, int act, float act_lo, float act_hi,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Single-channel MaskConv2d (Cin = 1, groups = 1, feature dilation 1, even feature stride; DS2's first convolution,
 * cnn.py:445-483 with builders/deep_speech_2.py:198-203 kernel [41, 11] stride [2, 2]) as the same split-bf16 implicit
 * GEMM: the KF input feature rows under an output feature row take the place of the input channels (padded to a
 * multiple of 16), the KT time taps stay taps.  x is [N, 1, Fin, Tin]; the workspace holds its feature-contiguous bf16
 * hi / lo planes [N, Tin, FP] with the SAME feature padding materialised; mask (t >= lens[n]) and time padding are load
 * predicates; bias + clamp in the epilogue; y is [N, Cout, Fout, Tout] float32.  MS_ERR_UNSUPPORTED when the staged
 * rows do not fit LDS (the caller then uses ms_maskconv_forward). */
size_t ms_maskconv_fwin_packed_bytes(int Cout, int KF, int KT);
int ms_maskconv_fwin_pack(const float* w, void* packed, int Cout, int KF, int KT, void* stream);
size_t ms_maskconv_fwin_workspace_bytes(int N, int Tin, int KF, int SF, int Fout);
int ms_maskconv_fwin_forward(const float* x, const int32_t* lens, const void* packed_w, const float* bias, float* y, int N,
                             int Fin, int Tin, int Cout, int Fout, int Tout, int KF, int KT, int SF, int ST, int DT,
                             int pad_f_l, int pad_t_l, int act, float act_lo, float act_hi, void* workspace,
                             size_t workspace_bytes, void* stream);

/* MaskConv1d with many input channels as im2col + split-bf16 GEMM (model/cnn.py:295-333, the conv1d flavour of the DS2
 * builder): x [N, Cin, Tin] -> y [N, Cout, Tout]; frames t >= lens[n] read as zero (cnn.py:280-293), pad_l zero
 * frames on the left (cnn.py:252-278); packed = ms_maskconv1d_gemm_pack of weight [Cout, Cin, KT]; bias may be NULL;
 * groups = 1.  Products carry the split-bf16 error (~2^-17 relative, like ms_linear_split_forward). */
size_t ms_maskconv1d_gemm_packed_bytes(int Cout, int Cin, int KT);
int ms_maskconv1d_gemm_pack(const float* w, void* packed, int Cout, int Cin, int KT, void* stream);
size_t ms_maskconv1d_gemm_workspace_bytes(int N, int Cin, int Tout, int Cout, int KT);
int ms_maskconv1d_gemm_forward(const float* x, const int32_t* lens, const void* packed, const float* bias, float* y, int N,
                               int Cin, int Tin, int Cout, int Tout, int KT, int ST, int DT, int pad_l, int act,
                               float act_lo, float act_hi, void* workspace, size_t workspace_bytes, void* stream);

/* DeepSpeech2._conv_to_rnn_size (deep_speech_2.py:114-117): [N, CF, T] -> [T, N, CF]. */
int ms_nct_to_tnc(const float* x, float* y, int N, int CF, int T, void* stream);

/* Elementwise clamp (SeqLenWrapper(Hardtanh/ReLU) on its own); y may alias x. */
int ms_clamp(const float* x, float* y, size_t n, float lo, float hi, void* stream);

/* ---- model/fully_connected.py ------------------------------------------ */

/* torch.nn.Linear (+ hidden activation) (fully_connected.py:107-131, 164; also
 * deep_speech_1.py:124-136): y[M,N] = act(x[M,K] . w[N,K]^T + bias[N]). bias may be NULL. */
int ms_linear_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                      float act_lo, float act_hi, void* stream);

/* Same contract as ms_linear_forward (exact float32 MFMA) for layers with few output COLUMNS (N <= 64, K >= 512: the
 * 29-symbol output layer, fully_connected.py:164 / deep_speech_1.py:118-120): one 32-column tile per 128 rows leaves a
 * streaming chunk's 1 024 rows on eight workgroups that each walk all of K, so the contraction is cut into min(8, K / 128)
 * slices whose partial sums land in the workspace and are added in slice order (deterministic; the slice count depends on
 * (K, N, flags) only -- never on M -- so a row's result does not depend on the batch or chunk it is in; the rounding differs
 * from ms_linear_forward's single k-ordered chain by a few ulp).  flags: 0, or MS_LINEAR_FEW_ROWS = the caller states that
 * the layer serves a handful of rows (one clip's hidden layers, deep_speech_1.py:124-136) and takes K slices for N <= 4096
 * too; a caller that sets it for some batches and not for others gives up bit-reproducibility between them.
 * ms_linear_splitk_workspace_bytes() == 0 means the shape is not such a layer: the call then IS ms_linear_forward and needs
 * no workspace; a NULL workspace selects ms_linear_forward for any shape. */
#define MS_LINEAR_FEW_ROWS 1
size_t ms_linear_splitk_workspace_bytes(int M, int K, int N, int flags);
int ms_linear_splitk_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                             float act_lo, float act_hi, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* Same contract as ms_linear_forward for K % 32 == 0, computed with float32 operands split
 * into bf16 hi + lo (x.w ~= x_hi.w_hi + x_lo.w_hi + x_hi.w_lo, f32 accumulate; relative error
 * ~2^-17 per product).  The workspace holds the four bf16 planes. */
size_t ms_linear_split_workspace_bytes(int M, int K, int N);
int ms_linear_split_forward(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int act,
                            float act_lo, float act_hi, void* workspace, size_t workspace_bytes, void* stream);

/* The same layer with its weight planes made once instead of per call (fully_connected.py:107-131 builds the Linear once and
 * calls it per batch): ms_linear_split_pack writes [hi | lo] planes of w [N, K] (ms_linear_split_packed_bytes(K, N) bytes, in
 * the precision mode of the process); ms_linear_split_forward_packed takes them, its workspace holds the x planes only
 * (M * K * 4 bytes, 256-byte aligned size).  Same kernels, same plane values, same bits as ms_linear_split_forward. */
size_t ms_linear_split_packed_bytes(int K, int N);
int ms_linear_split_pack(const float* w, void* packed, int K, int N, void* stream);
int ms_linear_split_forward_packed(const float* x, const void* packed_w, const float* bias, float* y, int M, int K, int N, int act,
                                   float act_lo, float act_hi, void* workspace, size_t workspace_bytes, void* stream);

/* ---- model/lookahead.py ------------------------------------------------- */

/* Lookahead.forward (lookahead.py:65-69): y[n,f,t] = sum_k w[f,k] * x[n,f,t+k]
 * (zero beyond T).  Element strides let the caller pass the [T,N,F] RNN output
 * directly (deep_speech_2.py:119-121, 161-164).  w is [F, ctx]. */
int ms_lookahead_forward(const float* x, const float* w, float* y, int N, int F, int T, int ctx, long xs_n,
                         long xs_f, long xs_t, long ys_n, long ys_f, long ys_t, int act, float act_lo,
                         float act_hi, void* stream);

/* The same on a WINDOW of a stream (streaming with carried context, myrtlespeech_amd/streaming.py): x holds T_in frames,
 * only the first T_out <= T_in output frames are computed and written (taps beyond T_in read as zero, which a caller
 * allows only at the true end of the utterance); y has room for T_out frames. */
int ms_lookahead_window_forward(const float* x, const float* w, float* y, int N, int F, int T_in, int T_out, int ctx,
                                long xs_n, long xs_f, long xs_t, long ys_n, long ys_f, long ys_t, int act, float act_lo,
                                float act_hi, void* stream);

/* ---- model/rnn.py / model/hard_lstm.py ---------------------------------- */

/* One layer (all directions) of torch.nn.LSTM/GRU/RNN as RNN.forward drives it
 * (rnn.py:170-183: pack_padded_sequence -> rnn -> pad_packed_sequence) or of
 * HardLSTM (hard_lstm.py:346-379, 416-456, 513-561).
 *
 * Packing: weights in torch layout, per direction d (0 fwd, 1 reverse):
 * w_ih[d] [G*H, In], w_hh[d] [G*H, H], b_ih[d]/b_hh[d] [G*H] or NULL (bias=False);
 * G = 4 (LSTM, gate order i,f,g,o), 3 (GRU: r,z,n), 1 (RNN).  The pointer arrays
 * themselves are HOST arrays of device pointers. */
/* rnn.py:112-120 accepts any hidden_size (protos/rnn.proto:20); the persistent recurrence kernels exist for LSTM widths that
 * are multiples of 64 (<= 1024) / 1280 / 1536 / 2048 and GRU widths 512 .. 2560 in steps of 256 (+ 1280).  Returns the width a
 * caller should PAD a layer of hidden size H to (zero weight rows / columns, zero biases, zero initial state for the padded
 * units: they stay exactly 0, so out[..., :H], h_n[..., :H], c_n[..., :H] are the unpadded layer's values with exact zeros
 * added to their sums) so that it runs on a persistent kernel instead of one launch per step; H itself when no padding is
 * needed or none helps.  Round 6: in the two-plane modes an LSTM width of 129 .. 1024 that is not 256 / 512 / 768 / 1024 is sent
 * to the NEXT of those (the two-stream kernel is about twice as fast as the one that serves the other multiples of 64), and a
 * width beyond every persistent kernel to the next multiple of 64 (the MFMA step kernel instead of the scalar one).
 * MS_RNN_PAD_HIDDEN=0 always returns H. */
int ms_rnn_padded_hidden(int cell, int H, int ndir);
size_t ms_rnn_packed_bytes(int cell, int In, int H, int ndir);
int ms_rnn_pack(int cell, int In, int H, int ndir, const float* const* w_ih_host, const float* const* w_hh_host,
                const float* const* b_ih_host, const float* const* b_hh_host, void* packed, void* stream);

size_t ms_rnn_workspace_bytes(int cell, int T, int N, int In, int H, int ndir);

/* x [T, N, In] time-major; lens [N] sorted descending (enforce_sorted=True,
 * rnn.py:174) or NULL = all T (HardLSTM ignores lengths, hard_lstm.py:36);
 * max_len = lens[0] (host copy; T if lens is NULL); h0/c0 [ndir, N, H] (c0 only
 * for LSTM cells; NULL = zeros, rnn.py:187-205); out [T, N, ndir*H] (rows
 * t >= lens[n] are written as 0); hn/cn [ndir, N, H] = state at each sequence's
 * last valid step. */
int ms_rnn_layer_forward(int cell, const void* packed, const float* x, const int32_t* lens, int max_len,
                         const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N, int In,
                         int H, int ndir, void* workspace, size_t workspace_bytes, void* stream);

/* Layer stacks (rnn.py:112-120 num_layers > 1): the same call with two optional hand-offs through the shared workspace,
 * for layer kinds where ms_rnn_layer_chains_planes() returns 1 (the two-stream LSTM with split-bf16 / fp16 operands, the persistent GRU):
 *   MS_RNN_OUT_PLANES_TO_WS  the layer output is left in the workspace as the NEXT layer's GEMM operand planes
 *                            ([max_len*N][ndir*H] bf16 hi + lo, or one fp16 plane) instead of float32 `out` (which may
 *                            then be NULL and is not written); the workspace must also hold
 *                            ms_rnn_workspace_bytes(cell, T, N, ndir*H, H, ndir);
 *   MS_RNN_X_PLANES_IN_WS    the input is taken from those planes (left there by the previous layer's call with the
 *                            same T, N, max_len and workspace); `x` may be NULL.
 * Same arithmetic as splitting the float32 output afterwards, one pass over the activations less per layer. */
enum { MS_RNN_X_PLANES_IN_WS = 1, MS_RNN_OUT_PLANES_TO_WS = 2, MS_RNN_PACKED_ROWS = 4, MS_RNN_HX_PREINIT = 4096,
       /* A layer call in two halves, for callers that interleave two batches' layers on two streams (model/rnn.py: the
        * half-batch pipeline of short sequences): MS_RNN_PROJECTION_ONLY = the input projection into the workspace, nothing
        * else; MS_RNN_RECURRENCE_ONLY = the recurrence on the projection an earlier MS_RNN_PROJECTION_ONLY call with the same
        * arguments left in the SAME workspace (stream-ordered behind it).  Together they do what the plain call does. */
       MS_RNN_RECURRENCE_ONLY = 8192, MS_RNN_TIMING_SKIP_PROJECTION = 8192 /* (tools/overlap_emulation.py's name for it) */,
       MS_RNN_PROJECTION_ONLY = 16384 };
int ms_rnn_layer_chains_planes(int cell, int H, int ndir);
/* One initialisation of the cross-workgroup exchange for a whole stack (rnn.py:112-120 num_layers > 1): every layer call
 * of the persistent kernels starts with a small launch that sets its exchange buffer's epoch tags and zeroes the per-call
 * status words -- five of them on a streaming chunk's critical path.  A workspace holds 8 exchange regions; a layer call
 * selects one with `flags` bits 8 .. 11 (0 by default).  ms_rnn_hx_preinit initialises regions 0 .. nregions - 1 for layers
 * that all run with this (cell, T, N, H, ndir, max_len) in ONE launch; the layer calls that follow on the same stream then
 * carry MS_RNN_HX_PREINIT | (region << 8), region = the layer's index.  MS_ERR_UNSUPPORTED (nothing launched, no error text)
 * for layer kinds it does not serve: the caller then simply leaves the flag out. */
int ms_rnn_hx_preinit(int cell, int T, int N, int In, int H, int ndir, int max_len, int nregions, void* workspace,
                      size_t workspace_bytes, void* stream);
/* MS_RNN_PACKED_ROWS (a permission, for batches whose lengths differ): torch's packed sequences (rnn.py:174-181) hold only
 * the frames t < lens[n]; with this flag the layer does the same where it can -- the input projection runs over sum(lens)
 * rows instead of max_len * N and the chained planes hold only those rows, frame after frame -- and ignores it where it
 * cannot.  Outputs are the same bits either way.  Whether a layer can depends on (cell, max_len, N, H, ndir), not on In:
 * ms_rnn_layer_packs_rows() says; every layer of a chained stack must be given the same flag. */
int ms_rnn_layer_packs_rows(int cell, int T, int N, int In, int H, int ndir);
int ms_rnn_layer_forward_ex(int cell, const void* packed, const float* x, const int32_t* lens, int max_len,
                            const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N, int In,
                            int H, int ndir, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* Synchronises `stream` and reports whether every ms_rnn_layer_forward that used
 * `workspace` since the previous call of this function completed (MS_OK) or a
 * persistent kernel timed out (MS_ERR_TIMEOUT, reported once).  The time-out word is
 * sticky across layer calls -- one check after a whole stack of layers is enough --
 * and lives in the first 256 bytes of the workspace, which must therefore be ZERO
 * when a freshly allocated workspace is used for the first time. */
/* 1 when a recurrent layer of this kind and batch size runs as the wide-workgroup persistent LSTM kernel, which occupies
 * at most half of the CUs per batch group of 32 rows (H = 1024, split-bf16 operands, up to 64 sequences): the two-in-flight
 * pipeline then runs the OTHER batch's projection as the regular GEMM on the free CUs instead of the co-tenant form. */
int ms_rnn_layer_is_wide(int cell, int H, int ndir, int N);
/* A whole stack in one call with layer l+1's input projection computed BESIDE layer l's recurrence (rnn.py:112-120 with
 * num_layers > 1, bidirectional; deep_speech_2.py:159): a bidirectional stack of wide-workgroup LSTM layers (H = 1024, up to 32
 * sequences) leaves half of the CUs idle during every recurrence; here a layer's recurrence runs as `segments` launches over
 * consecutive time segments and, after each, a second (library-owned) stream computes the next layer's two K-half projections
 * of the rows that segment produced.  Same kernels on the same operands as nl calls of ms_rnn_layer_forward_ex with the planes
 * chained -- the same bits -- for rows that all exist (not MS_RNN_PACKED_ROWS).  packed_host: HOST array of the nl layers'
 * packed weights (device pointers, layer 0 first; layer 0 packed for In, the others for ndir * H); h0 / c0 / hn / cn:
 * [nl * ndir, N, H]; out: [T, N, ndir * H] float32 of the last layer.  ms_rnn_stack_overlap_ok says whether a stack qualifies
 * (MS_RNN_OVERLAP=0: never).  Not for use inside a stream capture or beside another stream's persistent launches. */
int ms_rnn_stack_overlap_ok(int cell, int T, int N, int In, int H, int ndir, int nl);
int ms_rnn_stack_forward(int cell, const void* const* packed_host, const float* x, const int32_t* lens, int max_len,
                         const float* h0, const float* c0, float* out, float* hn, float* cn, int T, int N, int In, int H, int ndir,
                         int nl, int segments, void* workspace, size_t workspace_bytes, void* stream);

int ms_rnn_status(const void* workspace, void* stream);

/* Diagnostic: byte offset inside the RNN workspace of the per-workgroup stamp sums
 * [ndir*H/8][8] (u64 wall-clock ticks, 100 MHz: wait, MFMA loop, reduce+cell, publish)
 * that the persistent LSTM kernel fills when MS_LSTM_STAMPS=1 is set in the environment. */
size_t ms_rnn_debug_offset(int cell, int T, int N, int In, int H, int ndir);

/* Tuning switch for same-process A/B runs of the split-operand GEMM (tools/gemm_probe.py): 0 = the shipped choice,
 * 2 = the register-staged kernel (the round-1 kernel; bit-identical results).  Not part of the reference surface. */
int ms_gemm_set_variant(int variant);

/* Optional launch timing for bench.py's roofline line and its per-stage breakdown (not part of the reference
 * surface).  While enabled, the entry points on a DeepSpeech step bracket their launches with HIP events on the
 * caller's stream.  ms_prof_read synchronises those events and returns the summed milliseconds and launch counts
 * since the last read, MS_PROF_KINDS entries each, indexed by MS_PROF_*:
 * PROJECTION = input projection of one recurrent layer (operand split, where the producer did not hand planes
 * over, + GEMM), RECURRENCE = recurrent kernel(s) of one layer, GEMM_K_LARGE = the split-operand projection GEMM
 * kernel alone at In >= 1024, GEMM_K_SMALL = the same at In < 1024 (the first layer of a stack) -- both nested
 * inside PROJECTION --, CONV = one masked convolution call (mask, layout pass and kernel), LAYOUT = ms_nct_to_tnc,
 * LINEAR = one ms_linear(_split)_forward call, GREEDY = ms_ctc_greedy_decode, OTHER = clamp / mask / lookahead. */
enum {
  MS_PROF_PROJECTION = 0, MS_PROF_RECURRENCE = 1, MS_PROF_GEMM_K_LARGE = 2, MS_PROF_GEMM_K_SMALL = 3, MS_PROF_CONV = 4,
  MS_PROF_LAYOUT = 5, MS_PROF_LINEAR = 6, MS_PROF_GREEDY = 7, MS_PROF_OTHER = 8
};
#define MS_PROF_KINDS 9
int ms_prof_enable(int on);
int ms_prof_read(float* out_ms_host, int* out_n_host);

/* Diagnostic (tools/clock_probe.py; not part of the reference surface): ONE wave that samples the shader clock counter
 * and the 100 MHz wall clock `samples` times, about `spacing_us` apart, into out_dev[2 * samples] (u64 pairs
 * {wall ticks, shader cycles}).  Launched on a stream of its own beside other kernels it shows the clock the chip holds
 * under their load (MI355X_MICROARCH.md, DVFS give-back item 6).  samples * spacing_us is capped at 50 ms. */
int ms_clock_probe(unsigned long long* out_dev, int samples, int spacing_us, void* stream);

/* Diagnostic (bench.py's floors for the scan kernels; not part of the reference surface): one workgroup of `threads`
 * runs `steps` rounds of {LDS write, barrier, neighbour read, barrier}; out_dev[0] = elapsed ticks of the 100 MHz wall
 * clock (2 * steps barrier-separated phases), out_dev[1] = a checksum. */
int ms_barrier_chain_probe(unsigned long long* out_dev, int steps, int threads, void* stream);

/* ---- loss/ctc_loss.py ---------------------------------------------------- */

size_t ms_ctc_loss_workspace_bytes(int T, int N, int V, int S_max);

/* The first 256 bytes of a CTC workspace (forward or backward) hold a sticky time-out word; the caller provides it zeroed
 * before the first call and leaves it alone afterwards.  The four-wave alpha pipeline hands two values per frame from wave
 * to wave through an LDS mailbox with a bounded (0.5 s) spin; should an entry never arrive -- it cannot while every wave of
 * the workgroup runs -- the utterance's loss (and its gradient) is NaN AND the word is set.  ms_ctc_status synchronises
 * `stream`, returns MS_ERR_TIMEOUT once if the word was set and clears it (the CTC twin of ms_rnn_status; ctc_loss.py has
 * no such state: torch's kernel cannot time out).  A NaN loss WITHOUT the status is data: a frame whose log-softmax
 * normaliser is not finite (a NaN or +inf logit, a row of -inf) gives NaN as torch.nn.CTCLoss does, also under
 * zero_infinity. */
int ms_ctc_status(const void* workspace, void* stream);

/* CTCLoss.forward (ctc_loss.py:95-101) = LogSoftmax(dim=-1) + torch.nn.CTCLoss:
 * log-space alpha recursion per utterance.  logits [T,N,V] unnormalised;
 * targets int32, utterance n's labels start at tgt_offsets[n] (covers both the
 * padded [N,S] and the concatenated 1-D form, ctc_loss.py:74-83).
 * nll [N] = -log p(target | input) (reduction='none');
 * reduced [1] = 'sum' (reduction=2) or 'mean' (reduction=1: nll/clamp(len,1),
 * batch mean, ctc_loss.py:17-19); zero_infinity replaces inf by 0. */
#define MS_CTC_LOG_PROBS_IN 2 /* OR-ed into `zero_infinity`: `logits` already hold the values torch.nn.CTCLoss is to take as
                              * log-probabilities -- CTCLoss(dim != -1), ctc_loss.py:37-45, normalises over another axis
                              * (ms_log_softmax_axis) -- so no log-softmax over the symbols is applied */
int ms_ctc_loss_forward(const float* logits, const int32_t* in_lens, const int32_t* targets,
                        const int32_t* tgt_offsets, const int32_t* tgt_lens, float* nll, float* reduced, int T,
                        int N, int V, int S_max, int blank, int reduction, int zero_infinity, void* workspace,
                        size_t workspace_bytes, void* stream);

/* LogSoftmax over an axis other than the last one (ctc_loss.py:45 passes the constructor's `dim` through): x, y are
 * contiguous [outer, axis, inner]; y = x - logsumexp over `axis`. */
int ms_log_softmax_axis(const float* x, float* y, int outer, int axis, int inner, void* stream);
/* Its backward (autograd through LogSoftmax(dim), ctc_loss.py:45): y = the log-probabilities ms_log_softmax_axis wrote,
 * g = the gradient with respect to them; gx = g - exp(y) * sum over `axis` of g.  gx may alias g. */
int ms_log_softmax_axis_backward(const float* y, const float* g, float* gx, int outer, int axis, int inner, void* stream);

/* Gradient of ms_ctc_loss_forward's per-utterance losses with respect to the logits (what autograd gives the
 * reference through LogSoftmax + torch.nn.CTCLoss, loss/ctc_loss.py:95-101): alpha rows forward, beta rows backward,
 * grad_logits[t,n,k] = grad_nll[n] * (softmax(x)[t,k] - exp(logsumexp_{s: l'_s=k}(alpha_t(s)+beta_t(s)) + nll[n] - lp[t,k]))
 * for t < in_lens[n], 0 on padding frames and, with zero_infinity, for utterances whose loss is infinite.  grad_nll
 * [N] is the upstream gradient of each nll[n] (the host folds the reduction in: 1 for 'sum', 1/(N*max(len,1)) for
 * 'mean').  grad_logits [T,N,V] is fully written.  `zero_infinity` is the same bit field as in ms_ctc_loss_forward: with
 * MS_CTC_LOG_PROBS_IN the `logits` are taken as log-probabilities (no softmax over the symbols: lp = logits) and the result
 * is what torch.nn.CTCLoss's backward hands to the LogSoftmax(dim) in front of it, exp(lp) - exp(... - lp) (the caller
 * chains ms_log_softmax_axis_backward); values above 3 are rejected. */
size_t ms_ctc_loss_backward_workspace_bytes(int T, int N, int V, int S_max);
int ms_ctc_loss_backward(const float* logits, const int32_t* in_lens, const int32_t* targets, const int32_t* tgt_offsets,
                         const int32_t* tgt_lens, const float* grad_nll, float* grad_logits, int T, int N, int V,
                         int S_max, int blank, int zero_infinity, void* workspace, size_t workspace_bytes, void* stream);

/* ---- post_process/ctc_greedy_decoder.py ---------------------------------- */

/* CTCGreedyDecoder.forward (ctc_greedy_decoder.py:74-92): argmax over symbols
 * (ties -> lowest index), drop repeats and blanks.  x [T,N,V] (any real scores);
 * out_idx [N, T] int32, out_len [N]; row n's entries past out_len[n] are unspecified (alphabets beyond 64 symbols keep
 * the frames' arg maxes there before the in-place compaction). */
int ms_ctc_greedy_decode(const float* x, const int32_t* lens, int32_t* out_idx, int32_t* out_len, int T, int N,
                         int V, int blank, void* stream);

/* ---- post_process/ctc_beam_decoder.py ------------------------------------ */

size_t ms_ctc_beam_workspace_bytes(int T, int N, int V, int beam_width);

/* CTCBeamDecoder.forward (ctc_beam_decoder.py:175-258): prefix beam search in
 * linear float32 arithmetic in the reference's visiting order, bit-exact beam.
 * probs [T,N,V] normalised.  separator < 0 = None; word_factor [T+2] (or NULL
 * when separator < 0) holds float32((1 + n_words) ** word_weight) for
 * n_words = 0..T+1, evaluated by the host exactly as ctc_beam_decoder.py:248-253.  One call advances every
 * utterance over time steps [t_begin, t_end); state persists in `workspace`
 * (t_begin = 0 initialises it), so a host language model can be consulted
 * between steps: lm_factor [N, beam_width] float32 (or NULL) multiplies the
 * separator extension of beam entry w (= float32(lm(l+sep) ** lm_weight),
 * ctc_beam_decoder.py:222-228).  When finish != 0 the best prefix of each
 * utterance is written to out_idx [N, T] / out_len [N]; beam_len [N] and
 * beam_idx [N, beam_width, T] / beam_plen [N, beam_width] (may be NULL) expose
 * the current beam for the LM callback. */
int ms_ctc_beam_decode(const float* probs, const int32_t* lens, int32_t* out_idx, int32_t* out_len, int T, int N,
                       int V, int blank, int beam_width, float prune_threshold, int separator, const float* word_factor,
                       int t_begin, int t_end, const float* lm_factor, int finish, int32_t* beam_len,
                       int32_t* beam_idx, int32_t* beam_plen, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- feature front-end: data/preprocess.py + builders/pre_process_step.py ---- */

/* torchaudio.transforms.MFCC as built by builders/pre_process_step.py:33-43 (torchaudio==0.4.0,
 * environment.yml:171; absent here, so the pipeline is restated from its published algorithm):
 * centred, reflect-padded STFT with `window` [n_fft] -> power -> mel_fb [n_mels, n_fft/2+1] ->
 * 10*log10(max(.,1e-10)) floored at (per-utterance max - top_db) -> dct [n_mfcc, n_mels].
 * wave [N, max_samples] zero-padded, wave_lens [N] (each > n_fft/2); dft [2*(n_fft/2+1), n_fft] holds
 * the cos rows then the -sin rows.  out [N, n_mfcc, T], T = 1 + max_samples/hop; utterance n has
 * 1 + wave_lens[n]/hop frames, later frames are zero (what data/batch.py:7-42 pads with).
 * top_db < 0 disables the floor. */
size_t ms_mfcc_workspace_bytes(int N, int T, int n_fft, int n_mels, int n_mfcc);
int ms_mfcc_forward(const float* wave, const int32_t* wave_lens, const float* window, const float* dft,
                    const float* mel_fb, const float* dct, float* out, int N, int max_samples, int T, int n_fft,
                    int hop, int n_mels, int n_mfcc, float top_db, void* workspace, size_t workspace_bytes,
                    void* stream);

/* MFCCLegacy.__call__ (data/preprocess.py:262-319) = python_speech_features==0.6 `mfcc`
 * (environment.yml:169; absent here, restated from its published algorithm) in float64: samples
 * scaled to the int16 range and truncated, pre-emphasis, rectangular frames of frame_len every
 * frame_step samples (zero-padded tail), |DFT_nfft|^2/nfft, fbank [nfilt, nfft/2+1], log,
 * dct [numcep, nfilt], lifter [numcep], coefficient 0 := log(frame energy).  twiddle [nfft, 2] holds
 * (cos, sin)(2 pi j / nfft).  out [N, numcep, T] float32, T = frame count of max_samples; frames
 * past an utterance's own count are zero. */
int ms_mfcc_legacy_forward(const float* wave, const int32_t* wave_lens, const double* twiddle, const double* fbank,
                           const double* dct, const double* lifter, float* out, int N, int max_samples, int T,
                           int frame_len, int frame_step, int nfft, int nfilt, int numcep, double preemph,
                           void* stream);

/* Standardize.__call__ (data/preprocess.py:57-63) per utterance: y = (x - mean) / std (unbiased)
 * over x[n, :, t < lens[n]]; x, y [N, inner, T]; frames t >= lens[n] are written as 0.  lens may be
 * NULL (every utterance is T frames long).  y may alias x. */
size_t ms_standardize_workspace_bytes(int N);
int ms_standardize_forward(const float* x, const int32_t* lens, float* y, int N, int inner, int T, void* workspace,
                           size_t workspace_bytes, void* stream);

/* AddContextFrames.__call__ (data/preprocess.py:117-141): x [N, F, T] -> y [N, 2c+1, F, T],
 * y[n,w,f,t] = x[n,f,t+w-c] when both t and t+w-c lie inside utterance n, else 0. */
int ms_context_frames_forward(const float* x, const int32_t* lens, float* y, int N, int F, int T, int n_context,
                              void* stream);

/* SpecAugment.__call__ (data/preprocess.py:193-218), the zeroing half: the host draws the bands in
 * the reference's order; f_bands [N, n_f, 2] / t_bands [N, n_t, 2] hold (start, width) int32 pairs
 * on the device.  x [N, C, F, T] is modified in place. */
int ms_spec_augment_(float* x, const int32_t* f_bands, const int32_t* t_bands, int N, int C, int F, int T, int n_f,
                     int n_t, void* stream);

/* ---- RNN-T decode step (own specification: the reference snapshot has no transducer,
 *      SURVEY 0.3 / 8 a15; see myrtlespeech_amd/model/rnnt.py) ------------------------------ */

/* out[r,:] = table[idx[r],:]  (prediction-network embedding; idx clamped to [0, V1)). */
int ms_embedding_forward(const float* table, const int32_t* idx, float* out, int R, int D, int V1, void* stream);

/* logp[r,:] = log_softmax(w_out . tanh(enc_p[enc_row[r],:] + pred_p[r,:]) + b_out) for R hypothesis rows;
 * enc_p [*, J] are the projected encoder frames, pred_p [R, J] the projected predictor outputs. */
int ms_rnnt_joint_forward(const float* enc_p, const int32_t* enc_row, const float* pred_p, const float* w_out,
                          const float* b_out, float* logp, int R, int J, int V1, void* stream);

/* Per row b of scores [B, C]: the k largest entries in descending order (ties -> lowest index);
 * -inf entries are never selected (index -1 is written when fewer than k finite candidates exist). */
int ms_rnnt_topk(const float* scores, int32_t* out_idx, float* out_val, int B, int C, int k, void* stream);

/* Whole-batch RNN-T decode on the device (own specification, oracle/rnnt_oracle.py): one call enqueues the
 * complete frame loop; nothing is read back until the caller synchronises.  enc_p [T*N, J] are the encoder frames
 * after RNNTJoint.enc_proj (row t*N + n); the prediction network is Embedding(V+1, D) + L LSTM layers given in
 * torch layout (w_ih[l] [4H, In_l], w_hh[l] [4H, H], b_ih[l] / b_hh[l] [4H] or NULL, gate order i,f,g,o), w_pred
 * [J, H] (no bias), w_out [V+1, J], b_out [V+1]; blank = V.
 * greedy != 0: per frame, emit argmax labels until blank, at most max_symbols per frame; out_idx [N, T*max_symbols].
 * greedy == 0: time-synchronous beam search of width beam_width (<= 32) with max_symbols rounds per frame (blank
 * transitions of equal prefixes merged by float32 logaddexp evaluated in float64; the beam_width best label
 * extensions, ties to the lowest (hypothesis, label) index, stay live); out_idx [N, T*(max_symbols-1) + 1].
 * out_len [N]; out_score [N] (beam only, may be NULL) is the log-probability of the returned hypothesis.
 * The greedy decode is the one entry point besides ms_rnn_status that BLOCKS the host: the number of iterations depends on
 * the labels emitted, so the host polls a device counter a few iterations behind the launch queue and returns once the last
 * poll's copy has landed (the beam decode only enqueues).  MS_ERR_HIP if the pinned counter ring cannot be created. */
size_t ms_rnnt_decode_workspace_bytes(int T, int N, int V, int D, int H, int L, int J, int beam_width, int max_symbols,
                                      int greedy);
int ms_rnnt_decode(const float* enc_p, const int32_t* lens, const float* embedding, const float* const* w_ih,
                   const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, const float* w_pred,
                   const float* w_out, const float* b_out, int32_t* out_idx, int32_t* out_len, float* out_score, int T,
                   int N, int V, int D, int H, int L, int J, int beam_width, int max_symbols, int greedy, void* workspace,
                   size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MS_HOTPATH_H */
