/* lvae.h -- C ABI of liblvae_hip.so: the MI355X (gfx950) kernels behind the aggressive-VAE training hot path.
 *
 * The reference (jxhe/vae-lagging-encoder) has no FFI layer: its hot path calls PyTorch/ATen ops from Python
 * (SURVEY.md 2.2, 8b).  Each entry point below replaces one implicit ATen op (or a fused group of them) at the
 * reference call site cited next to it.  Conventions for every function:
 *   - plain device pointers + sizes + a hipStream_t passed as void*; no torch types, no C++ types, no exceptions;
 *   - return int: 0 = ok, > 0 = hipError_t of the launch, < 0 = argument check (-1 arg, -2 shape, -3 alignment,
 *     -4 unsupported);
 *   - the caller owns every buffer including workspaces; functions never allocate, free or synchronise, and are
 *     stream-ordered on `stream` (graph-capture safe); re-entrant, no global mutable state;
 *   - scalars that change between hipGraph replays (kl weight, lr, clip coefficient, Adam step) are read from
 *     device memory (`*_dev` arguments).
 * Layouts: token ids int64 batch-first [B][ids_stride] (as data/text_data.py:219-255 builds them); activations
 * time-major, row r = t*B + b; dropout keep-masks uint8 batch-first [B][T][C] (the reference's tensor layout).
 */
#ifndef LVAE_H
#define LVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense contractions (exact f32 on v_mfma_f32_32x32x2_f32) ------------------------------------------------
 * C[M,N](ldc) = alpha * op(A)[M,K] . op(B)[K,N] (+ add1[(row % mod1)*ld1 + col]) (+ add2[...]) (+ C if accumulate)
 * transA = 0: A stored [M][K];  1: stored [K][M].   transB = 0: B stored [K][N];  1: stored [N][K].
 * Replaces aten::mm/addmm under nn.LSTM's input projection (modules/encoders/enc_lstm.py:60,
 * modules/decoders/dec_lstm.py:104), nn.Linear (enc_lstm.py:62; dec_lstm.py:99,109) and their backward
 * (text.py:384).  The add1/add2 epilogue carries the two LSTM biases and the decoder's z-projection, so
 * torch.cat((word_embed, z_), -1) (dec_lstm.py:97) is never materialised. */
int lv_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                const float* A, long lda, const float* B, long ldb, float* C, long ldc, int accumulate,
                const float* add1, long ld1, int mod1, const float* add2, long ld2, int mod2,
                float* ws /* optional split-K scratch */, long ws_floats, void* stream);

/* Same contract on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16): f32 operands in HBM are rounded to bf16 while
 * being staged, accumulation and outputs are f32.  The throughput configuration of BASELINE.json (bf16). */
int lv_gemm_bf16(int transA, int transB, int M, int N, int K, float alpha,
                 const float* A, long lda, const float* B, long ldb, float* C, long ldc, int accumulate,
                 const float* add1, long ld1, int mod1, const float* add2, long ld2, int mod2,
                 float* ws, long ws_floats, void* stream);

/* bf16 matrix pipe over operands that are already bf16 in HBM (raw bits, uint16_t): the vocabulary-sized
 * contractions of LSTMDecoder (pred_linear forward dec_lstm.py:117 and its two backward products).  B is stored
 * [N][K]; A is stored [M][K] (transA = 0) or [K][M] (transA = 1); lda % 8 == 0, ldb % 8 == 0, both 16 B-aligned
 * (else LV_ERR_ALIGN).  Same epilogue / split-K contract as lv_gemm_f32.  Operands
 * rounded with lv_cvt_bf16_f32 / lv_softmax_nll_bwd_b16 give results bit-identical to lv_gemm_bf16 on the f32 data. */
int lv_gemm_b16(int transA, int M, int N, int K, float alpha,
                const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C, long ldc, int accumulate,
                const float* add1, long ld1, int mod1, const float* add2, long ld2, int mod2,
                float* ws, long ws_floats, void* stream);
/* LSTMDecoder's vocabulary projection fused with the statistics of nn.CrossEntropyLoss (modules/decoders/dec_lstm.py:117,
 * 140-146): logits = A . B^T (operands as lv_gemm_b16, transA = 0) written once as IEEE binary16 [M][ldl16] (RNE of the f32
 * accumulators; the f32 logits image is never produced) + per-row statistics over 64-column pieces, part [M][parts] (max,
 * sum exp(x - max)) with parts = lv_gemm_b16_nll_parts(N), + tgt_logit [M] = the logit of row r's target token
 * ids[(r % Bsz) * ids_stride + r / Bsz + tgt_off].  lv_softmax_nll_merge_f32 finishes lse / nll; lv_softmax_nll_bwd_h16 is the
 * backward over the binary16 image.  ldl16 % 8 == 0. */
/* ONE product, TWO destinations: C1 [M][nsplit] = columns [0, nsplit) of op(A) . B^T, C2 [M][N - nsplit] the rest (lv_gemm_b16 otherwise:
 * bf16 operand images, f32 accumulation; no addends).  The two weight gradients of an LSTM layer that share their A operand --
 * dW_ih = dG^T X and dW_hh = dG^T h_prev, the backward of nn.LSTM at enc_lstm.py:55 / dec_lstm.py:104 -- as one launch over the image
 * [X^T ; h_prev^T] ([ni + H][T*B]) and one reduction stage.  lv_gemm_b16_dual_supported: 1 where the shape takes the split-K route
 * the column split rides in (not the 256 x 256 tile), else 0 and the entry returns LV_ERR_UNSUPPORTED. */
int lv_gemm_b16_dual_supported(int M, int N, int K, long ws_floats);
int lv_gemm_b16_dual(int transA, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                     float* C1, long ldc1, int nsplit, float* C2, long ldc2, float* ws, long ws_floats, void* stream);
/* TWO independent products in ONE launch of one workgroup per CU: C0 (| C0b) = op(A0) . B0^T and C1 = A1 . B1^T (operands, layouts and
 * alignment as lv_gemm_b16; plain outputs; nsplit0 > 0: columns >= nsplit0 of the first product go to C0b as in lv_gemm_b16_dual; the
 * second product is in the transA = 0 form; M1 == 0: one product).  The backward of one nn.LSTM layer's input side (enc_lstm.py:55 /
 * dec_lstm.py:104 under autograd): [dW_ih | dW_hh] = dG^T [X ; h_prev] and dX = dG . W_ih, which as single launches leave 256 x 256 tiles
 * for 96 and 50 of the 256 CUs.  Both products run on the 256 x 256 quadrant K loop; a tile whose K range is shared between workgroups
 * is summed INSIDE the launch, in descending K order whoever does it (deterministic): by the workgroup that closes the tile's K range if
 * it finds the others arrived (a bounded look; its own piece then never leaves its registers), else by whichever draws the last ticket
 * (no reduction launch, no wait that progress depends on).
 * lv_gemm_b16_pair_supported: 1 where that is possible AND worth it (both products present, >= 12 K tiles per workgroup), else 0; the
 * entry itself only refuses the impossible (LV_ERR_UNSUPPORTED: second product transposed, > 1024 tiles, ws_floats < 2 * 256 * 65536). */
int lv_gemm_b16_pair_supported(int transA0, int M0, int N0, int K0, int transA1, int M1, int N1, int K1, long ws_floats);
int lv_gemm_b16_pair_pending(int* pending, void* stream);   /* diagnostic: *pending (device int) = arrival counters not back at zero; 0 between launches */
int lv_gemm_b16_pair(int transA0, int M0, int N0, int K0, const uint16_t* A0, long lda0, const uint16_t* B0, long ldb0,
                     float* C0, long ldc0, int nsplit0, float* C0b, long ldc0b,
                     int transA1, int M1, int N1, int K1, const uint16_t* A1, long lda1, const uint16_t* B1, long ldb1,
                     float* C1, long ldc1, float* ws, long ws_floats, void* stream);
/* C [M][N] = (A . B^T) * (keep ? kscale : 0): lv_gemm_b16 (transA = 0, plain output, ldc = N) with the backward of nn.Dropout on the
 * LSTM output (dec_lstm.py:106; dO = dlogits . W_pred then masked) applied in the product's reduction stage instead of by a pass of
 * its own; rows time-major (r = t * Bsz + b), keep = the reference-layout mask [Bsz][M / Bsz][N], uint8. */
int lv_gemm_b16_keep(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C,
                     const uint8_t* keep, float kscale, int Bsz, float* ws, long ws_floats, void* stream);
/* C = op(A) . B^T as lv_gemm_b16 computes it (alpha 1, no addends) and, in the same pass, the product's sum of squares:
 * sq[0 .. parts) receives one partial per wave of the kernels that hold C's final values (fixed slots: deterministic), parts =
 * lv_gemm_b16_sumsq_parts(M, N, K, ws_floats); 0 parts = the shape does not take the 256 x 256 tile and lv_gemm_b16_sumsq refuses it
 * (LV_ERR_ARG).  sq_only != 0: C is NOT written -- a weight gradient that only enters the norm of clip_grad_norm_ (text.py:383-387:
 * the inner loop clips over all parameters and steps the encoder alone; dW_pred is such a gradient). */
int lv_gemm_b16_sumsq_parts(int M, int N, int K, long ws_floats);
int lv_gemm_b16_sumsq(int transA, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C, long ldc,
                      float* ws, long ws_floats, float* sq, int sq_only, void* stream);
int lv_gemm_b16_nll_parts(int N);
/* The same two entries with the tile edge named by the caller instead of chosen by shape: tile = 0 (by shape: the 256 x 256 x 64
 * kernel, one workgroup per CU, for products of >= 1e11 flop -- the three vocabulary-sized GEMMs of dec_lstm.py:117,140-146 --
 * else 128 x 128 x 64), 128 or 256 (tests, microbenchmarks; anything else LV_ERR_ARG).  A per-call argument: the library keeps no
 * process-wide state. */
int lv_gemm_b16_tile(int tile, int transA, int M, int N, int K, float alpha,
                     const uint16_t* A, long lda, const uint16_t* B, long ldb, float* C, long ldc, int accumulate,
                     const float* add1, long ld1, int mod1, const float* add2, long ld2, int mod2,
                     float* ws, long ws_floats, void* stream);
int lv_gemm_b16_nll_tile(int tile, int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, uint16_t* logits16,
                         long ldl16, const int64_t* ids, long ids_stride, int tgt_off, int Bsz, float* part, float* tgt_logit,
                         void* stream);
int lv_gemm_b16_nll(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, uint16_t* logits16, long ldl16,
                    const int64_t* ids, long ids_stride, int tgt_off, int Bsz, float* part, float* tgt_logit, void* stream);
int lv_softmax_nll_merge_f32(const float* part, int nparts, const float* tgt_logit, float* lse, float* nll, int R, void* stream);
int lv_softmax_nll_bwd_h16(const uint16_t* logits16, long ldl, const float* lse, const int64_t* ids, long ids_stride, int tgt_off,
                           const float* rowscale, uint16_t* dlogits, long ldo, int T, int B, int V, void* stream);
/* src f32 [R][C](lds) -> dst bf16 [R][C](ldd) and/or dstT bf16 [C][R](ldt), round-to-nearest-even; either may be NULL */
int lv_cvt_bf16_f32(const float* src, long lds, int R, int C, uint16_t* dst, long ldd, uint16_t* dstT, long ldt,
                    void* stream);
/* the same images with nn.Dropout folded in (dec_lstm.py:106, dropout_out on the decoder LSTM's output): src = h time-major
 * [T*Bsz][C] (row t*Bsz + b), keep = the reference-layout mask [Bsz][T][C] (uint8): images of h * (keep ? kscale : 0).  And its
 * backward on dO, in place.  (The persistent recurrences then run without a mask: +0.3 us per timestep otherwise.) */
int lv_cvt_bf16_keep_f32(const float* src, long lds, int T, int Bsz, int C, const uint8_t* keep, float kscale,
                         uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream);
/* dst f32 [n] = bf16 src [n] * scale: unpacks a bf16 gradient payload of the data-parallel exchange (dist.GradSync payload="bf16") */
int lv_cvt_f32_bf16_scaled(const uint16_t* src, long n, float scale, float* dst, void* stream);
int lv_keep_scale_f32(float* x, const uint8_t* keep, float kscale, int T, int Bsz, int C, void* stream);

/* LSTM gate weight W [4H][C] (rows g*H + u): dst = bf16 image with rows in unit-major order (4u + g), dstT = bf16 image
 * of W^T [C][4H] in the standard order; either may be NULL */
int lv_cvt_bf16_gates_f32(const float* src, long lds, int H, int C, uint16_t* dst, long ldd, uint16_t* dstT, long ldt,
                          void* stream);
/* The LOW half of a split-bf16 operand, bf16(x - bf16(x)), in the layouts of the three conversions above: a plain matrix [R][C]
 * (gate_H = 0, ids = NULL), an LSTM gate weight with unit-major rows (gate_H = H, R = 4H), or gathered embedding rows (ids [Bsz]
 * [ids_stride], R = T*Bsz).  x = hi + lo up to 2^-17 |x|; hi.hi' + hi.lo' + lo.hi' on the bf16 pipe (lv_gemm_b16 with
 * accumulate = 1) is an f32-like product.  Used by the encoder's exact forward (enc_lstm.py:47-64 -> encoder.py:55: the KL is a
 * function of the forward's last state, and WEIGHT rounding is what moves it -- profiles/r05a_kl_ablation.txt). */
int lv_cvt_bf16_lo_f32(const float* src, long lds, int R, int C, int gate_H, const int64_t* ids, long ids_stride, int Bsz, int V,
                       uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream);
/* Operand images for a forward product on the BINARY16 matrix pipe: dst = IEEE half (RNE) in the plain / unit-major gate rows / gathered
 * embedding rows layouts (arguments as lv_cvt_bf16_lo_f32), dstT = the transposed BF16 image the gradient products of the same operand
 * read.  lv_gemm_h16 = lv_gemm_b16 (transA = 0, 128 x 128 tile) on such operands: v_mfma_f32_32x32x16_f16, f32 accumulation.  The
 * encoder's input projection X W_ih^T (enc_lstm.py:50-55).  Values beyond +-65504 saturate. */
int lv_cvt_h16_f32(const float* src, long lds, int R, int C, int gate_H, const int64_t* ids, long ids_stride, int Bsz, int V,
                   uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream);
int lv_gemm_h16(int M, int N, int K, float alpha, const uint16_t* A, long lda, const uint16_t* B, long ldb,
                float* C, long ldc, int accumulate, const float* add1, long ld1, int mod1,
                const float* add2, long ld2, int mod2, float* ws, long ws_floats, void* stream);
/* out[cols][rows] = in[rows][cols]^T  (W_hh^T for BPTT) */
int lv_transpose_f32(const float* in, float* out, int rows, int cols, void* stream);
int lv_transpose_ld_f32(const float* in, long in_ld, float* out, long out_ld, int rows, int cols, void* stream);

/* ---- LSTM time recurrence: nn.LSTM forward (enc_lstm.py:60, dec_lstm.py:104) and its autograd backward --------
 * gx [T][B][4H] input projection (+biases, + z term); whh [4H][H] (rows i|f|g|o); hs, cs [T+1][B][H] with index 0
 * = initial state supplied by the caller; gates: T*B*4H floats of activated gates saved for BPTT (opaque: unit-major
 * (i,f,g,o) records, written by the forward and read back only by lv_lstm_bwd_*).
 * dmask (optional) uint8 [B][T][H] keep-mask of nn.Dropout on the outputs (dec_lstm.py:106): hdrop[t] =
 * hs[t+1]*mask*dscale;  with dmask == NULL and hdrop != NULL, hdrop is a copy of the outputs. */
int lv_lstm_fwd_f32(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                    const uint8_t* dmask, float dscale, float* hdrop, float* ws, int T, int B, int H, void* stream);
/* same recurrence with the h W_hh^T product on the bf16 matrix pipe (throughput configuration; f32 accumulate/state) */
int lv_lstm_fwd_bf16(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                     const uint8_t* dmask, float dscale, float* hdrop, float* ws, int T, int B, int H, void* stream);
/* lv_lstm_fwd_bf16 for a gx whose 4H columns are UNIT-major (column 4u + g instead of g*H + u): each workgroup then
 * fetches a unit's four gate pre-activations with one 16-byte load.  Such a gx is what lv_gemm_b16 produces from the
 * lv_cvt_bf16_gates_f32 image of W_ih and lv_gate_interleave_f32-ed epilogue addends. */
int lv_lstm_fwd_bf16_ug(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                        const uint8_t* dmask, float dscale, float* hdrop, float* ws, int T, int B, int H, void* stream);
/* the exact-f32 recurrence (lv_lstm_fwd_f32) on a UNIT-major gx: nn.LSTM of enc_lstm.py:55 in f32 behind an input projection in
 * the bf16-image path's column order */
int lv_lstm_fwd_f32_ug(const float* gx, const float* whh, float* hs, float* cs, float* gates,
                       const uint8_t* dmask, float dscale, float* hdrop, float* ws, int T, int B, int H, void* stream);
/* The same recurrences as ONE persistent launch each (lv_lstm_persist16.hip; nn.LSTM of enc_lstm.py:55 / dec_lstm.py:104, forward
 * and BPTT): 256 workgroups in 8 XCD-sized groups (blockIdx % 8), each group carries a slice of the batch through all T steps with
 * its slice of W_hh held in registers and hands h_t (forward: all-gather) or partial dh sums (BPTT: reduce-scatter) around inside
 * the group through tagged 16-byte granules (every dword under a tag of its own: no 16-byte atomicity assumed).  R rows per group (1 <= R <= 16, 8 R >= B), groups [0, ceil(B / R)) carry the batch and
 * the workgroups of the remaining groups return at once -- B = 32 runs 4 rows on each of the 8 groups, B = 128 sixteen (BASELINE.json
 * configs[4]), and R = 8 at B = 32 runs a recurrence on FOUR XCDs.  Contraction on the 16 x 16 x 32 MFMA with the weights as the A
 * operand.  gx unit-major as for lv_lstm_fwd_bf16_ug.  Weight images: lv_lstm_persist16_pack(whh, wpk, backward, H)
 * (lv_lstm_persist16_wpk_floats() floats each; a caller whose weights do not change between calls -- the decoder during the
 * aggressive inner loop, text.py:371-400 -- packs once); exchange buffer: lv_lstm_persist16_xch_floats() floats; *status (device
 * int, zeroed by the caller) becomes non-zero if a bounded hand-off spin ran out (see lv_clip_norm2_txn_f32 for what the fused
 * driver does about it).  No in-kernel
 * dropout (the caller applies dropout_out on the bf16 images of h and once on dO).  flags bit 0: the hand-off granules are stored
 * without the agent-scope write-through, i.e. they stay in the XCD's L2 instead of travelling to memory (valid while every group is
 * XCD-local, which the round-robin workgroup placement of a 256-CU device gives; a violated assumption shows up as a hand-off
 * timeout in *status).  LV_ERR_UNSUPPORTED unless H == 1024 and the device has >= 256 CUs.
 * saved: what the forward keeps for the BPTT (the gates i, f, g, o and the cell states c_0 .. c_{T-1}), in the kernels' own
 * WORKGROUP-MAJOR order -- saved[group][member][t]{ gates [R][32 units][4], c [R][32 units] }: lv_lstm_persist16_saved_floats(T, R)
 * floats, opaque to the caller, and the BPTT call must be given the T, B, R of the forward that wrote it.  (In [t][b][...] order
 * every timestep of every array is a different page at B = 128, and the address translations of a block of loads serialise in
 * front of the next hand-off poll: 1.5 us of a 7.9 us BPTT timestep.)  hs: [T + 1][B][H] as everywhere; cs: [T + 1][B][H], of which
 * these kernels read slot 0 (the initial state) and write slot T (the final one) only. */
long lv_lstm_persist16_wpk_floats(void);
/* exchange buffer [forward half 0 | forward half 1 | BPTT half 0 | BPTT half 1].  flags bit 1 (2) of the two launches: double-buffered
 * -- bit 2 (4) names the half this launch uses, which must be zero (allocate zeroed; alternate strictly per kind of launch); the launch
 * zeroes the OTHER half of its kind in its prologue, so no memset launch precedes it.  BPTT: bits 3..4 name the instantiation (1 / 2 / 3 =
 * up to 4 / 8 / 16 rows per group; 0 = this launch's own) of the launch that last used the other half = the extent to clear.  Without
 * bit 1: half 0 behind a hipMemsetAsync (hipGraph replays cannot alternate).  lv_lstm_persist16_xch_clear zeroes the whole buffer. */
long lv_lstm_persist16_xch_floats(void);
int lv_lstm_persist16_xch_clear(float* xch, void* stream);
long lv_lstm_persist16_saved_floats(int T, int R);
int lv_lstm_persist16_pack(const float* whh, float* wpk, int backward, int H, void* stream);
int lv_lstm_persist16_pack2(const float* whh, float* wpk_fwd, float* wpk_bwd, int H, void* stream);   /* both images, one launch */
/* BINARY16 recurrent operands for the FORWARD (flags bit 5 = 32 of lv_lstm_fwd_bf16_persist16; weight image from
 * lv_lstm_persist16_pack(.., backward = 2, ..) or lv_lstm_persist16_pack2_h16, which also writes the bf16 BPTT image): W_hh and the h
 * granules as IEEE half on v_mfma_f32_16x16x32_f16 -- 11 instead of 8 bits of significand at the same cost.  For the ENCODER
 * (enc_lstm.py:55-62): the KL of encoder.py:55 is a function of the forward's last state, and what moves that state in the bf16
 * configuration is the rounding of the WEIGHTS (the same perturbation at every timestep), not of h.  LSTM weights and h in (-1, 1) are
 * far inside binary16's range; the BPTT stays bf16 (gradients need the exponent range). */
int lv_lstm_persist16_pack2_h16(const float* whh, float* wpk_fwd, float* wpk_bwd, int H, void* stream);
int lv_lstm_fwd_bf16_persist16(const float* gx, const float* wpk, float* hs, float* cs, float* saved, float* xch, int* status,
                               int T, int B, int R, int flags, int H, void* stream);
int lv_lstm_bwd_bf16_persist16(const float* dh_ext, const float* dh_last, const float* wpk, const float* saved, const float* hs,
                               const float* cs, uint16_t* dG16, float* dGsum, float* xch, int* status, float* dh0, float* dc0,
                               int tanh_init, int T, int B, int R, int flags, int H, void* stream);
/* Saved activations of a forward that ran on the launch-per-timestep kernels (gates [T][B][H][4] records, cs [T+1][B][H]) -> the
 * record buffer lv_lstm_bwd_bf16_persist16 reads, for the same T, B, R.  Used when the ENCODER's forward recurrence runs in exact
 * f32 (the KL of encoder.py:55 is a function of its last state, enc_lstm.py:60-62) in front of the persistent BPTT. */
int lv_lstm_persist16_import_saved(const float* gates, const float* cs, float* saved, int T, int B, int R, int H, void* stream);
/* out[r][4u + g] = a[r][g*H + u] (+ b[r][g*H + u]): gate-major rows (biases, the decoder's z-projection) -> unit-major */
int lv_gate_interleave_f32(const float* a, const float* b, int R, int H, float* out, void* stream);
int lv_lstm_bwd_bf16(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                     const float* whh, const float* gates, const float* hs, const float* cs,
                     float* dG, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                     int T, int B, int H, void* stream);
/* lv_lstm_bwd_bf16 that also (or only) writes dG16, the bf16 image of dG ([T][B][4H], bits identical to
 * lv_cvt_bf16_f32 of dG) that lv_gemm_b16 consumes for dX / dW_ih / dW_hh; either of dG / dG16 may be NULL, not both. */
int lv_lstm_bwd_bf16_img(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                           const float* whh, const float* gates, const float* hs, const float* cs,
                           float* dG, uint16_t* dG16, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                           int T, int B, int H, void* stream);
/* floats of caller-owned scratch (16-byte aligned) both LSTM entry points need: MFMA-fragment-major packed copies of
 * W_hh and of the recurrent state, split-K slabs */
long lv_lstm_ws_floats(int B, int H);
/* number of split-K slabs the BPTT matmul writes per step (informational) */
int lv_lstm_bwd_ksplit(int H);
/* BPTT.  dh_ext (optional) [T][B][H] grad wrt every output (times dmask*dscale when dmask given); dh_last
 * (optional) [B][H] grad wrt the final output only (encoder: enc_lstm.py:60-62 uses last_state only).
 * Outputs: dG [T][B][4H] grad wrt pre-activations; dGsum [B][4H] = sum_t dG[t]; dc0/dh0 (optional) grads wrt the
 * initial state; tanh_init = 1 when h0 = tanh(c0) (dec_lstm.py:99-101): then dc0 includes the path through h0. */
int lv_lstm_bwd_f32(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                    const float* whh, const float* gates, const float* hs, const float* cs,
                    float* dG, float* dGsum, float* ws, float* dh0, float* dc0, int tanh_init,
                    int T, int B, int H, void* stream);

/* ---- embeddings: nn.Embedding forward (enc_lstm.py:58, dec_lstm.py:80) fused with dropout_in (dec_lstm.py:81);
 * aten::embedding_dense_backward as a deterministic sorted-segment sum (padding_idx row skipped: dec_lstm.py:28) */
int lv_embed_gather_f32(const float* emb, const int64_t* ids, long ids_stride, const uint8_t* mask, float scale,
                        float* X, int T, int B, int ni, int V, void* stream);
/* The same lookup written straight into the bf16 operand images lv_gemm_b16 consumes (throughput configuration): dst [T*Bsz][C]
 * (row t*Bsz + b) and/or dstT [C][T*Bsz]; keep = the dropout keep-mask [Bsz][T][C] or NULL.  Bit-identical to
 * lv_embed_gather_f32 + lv_cvt_bf16_f32 (enc_lstm.py:50-52, dec_lstm.py:86-93). */
int lv_embed_gather_b16(const float* emb, const int64_t* ids, long ids_stride, const uint8_t* keep, float kscale,
                        int T, int Bsz, int C, int V, uint16_t* dst, long ldd, uint16_t* dstT, long ldt, void* stream);
int lv_token_sort(const int64_t* ids, long ids_stride, int T, int B, int V,
                  int* out_rows, int* out_tok, int* tmp /* 2*T*B ints */, void* stream);
int lv_embed_scatter_f32(const float* dX, const uint8_t* mask, float scale, const int* sorted_rows,
                         const int* sorted_tok, int T, int B, float* dE, int ni, int pad_idx, int accumulate,
                         void* stream);
/* the same sums with dE [V][ni] COMPLETE: the rows of tokens that do not occur in the batch (and pad_idx) are written as zeros
 * by the same launch -- the embedding gradient of nn.Embedding (enc_lstm.py:33, dec_lstm.py:34) without a fill of the table */
int lv_embed_scatter_full_f32(const float* dX, const uint8_t* mask, float scale, const int* sorted_rows,
                              const int* sorted_tok, int T, int B, float* dE, int ni, int V, int pad_idx, void* stream);
/* lv_embed_scatter_full_f32 and, in the same pass, the sum of squares of the table gradient it completes: sq[0 .. parts) receives
 * one partial per wave (parts = lv_embed_scatter_sumsq_parts(T, B)); sq_only != 0: dE is NOT written (see lv_gemm_b16_sumsq). */
int lv_embed_scatter_sumsq_parts(int T, int B);
int lv_embed_scatter_full_sumsq_f32(const float* dX, const uint8_t* mask, float scale, const int* sorted_rows,
                                    const int* sorted_tok, int T, int B, float* dE, int ni, int V, int pad_idx, float* sq, int sq_only,
                                    void* stream);
/* Data parallel, row-list exchange of the embedding gradient (new design; the reference has no distributed code): the
 * dense MEAN gradient dE [V][ni] rebuilt from every rank's (sorted token ids, gradient rows) list -- ids_all int64 [world][cap]
 * (ascending, -1 padded), rows_all [world][cap][ni] f32 (b16 = 0) or bf16 (b16 = 1) --, rows added in rank order (deterministic),
 * times scale (1 / world); rows no rank touched are written as zeros. */
int lv_rows_merge_f32(const int64_t* ids_all, const void* rows_all, int b16, int world, int cap, int ni, int V, float scale,
                      float* dE, void* stream);

/* ---- reparameterise + analytic KL: GaussianEncoderBase.encode / reparameterize (modules/encoders/encoder.py:40-79)
 * mulv [B][2nz] = mu | logvar; eps [B][ns][nz] is an input (host RNG in parity mode, lv_rng_* otherwise). */
int lv_reparam_kl_fwd_f32(const float* mulv, const float* eps, float* z, float* kl, int B, int ns, int nz, void* stream);
int lv_reparam_kl_bwd_f32(const float* mulv, const float* eps, const float* dz, const float* dkl, float* dmulv,
                          int B, int ns, int nz, void* stream);

/* ---- token NLL: nn.CrossEntropyLoss(reduce=False) + view().sum(-1) (dec_lstm.py:45-47,143-148) ----------------
 * logits [T*B][ldl]; target of row (t,b) = ids[b*ids_stride + t + tgt_off].  bwd rewrites logits in place with
 * (softmax - onehot) * rowscale[b]. */
int lv_softmax_nll_fwd_f32(const float* logits, long ldl, const int64_t* ids, long ids_stride, int tgt_off,
                           float* lse, float* nll, int T, int B, int V, void* stream);
int lv_softmax_nll_bwd_f32(float* logits, long ldl, const float* lse, const int64_t* ids, long ids_stride, int tgt_off,
                           const float* rowscale, int T, int B, int V, void* stream);
/* same gradient written as bf16 into dlogits [T*B][ldo] (pad columns zeroed); logits are left untouched */
int lv_softmax_nll_bwd_b16(const float* logits, long ldl, const float* lse, const int64_t* ids, long ids_stride, int tgt_off,
                           const float* rowscale, uint16_t* dlogits, long ldo, int T, int B, int V, void* stream);
/* VAE.loss assembly (modules/vae.py:95-98): rec[b] = sum_t nll[t][b]; loss[b] = rec[b] + kl_weight*kl[b] */
int lv_vae_loss_f32(const float* nll, const float* kl, const float* kl_weight_dev, float* loss, float* rec,
                    int T, int B, void* stream);
/* upstream grads (each may be NULL) -> rowscale[b] = g_loss+g_rec, dkl[b] = kl_weight*g_loss+g_kl */
int lv_loss_bwd_scales_f32(const float* g_loss, const float* g_rec, const float* g_kl, const float* kl_weight_dev,
                           float* rowscale, float* dkl, int B, void* stream);
/* The fused driver's form of the three entries above plus the running report sums (text.py:381,426-427), one launch:
 * rec[b] = sum_t nll[t][b]; loss[b] = rec[b] + w*kl[b]; acc_dev[0..2] += sum_b (loss, rec, kl); rowscale[b] = g_loss[b];
 * dkl[b] = w*g_loss[b] (the seeds of loss.mean().backward(), text.py:382-384) */
int lv_loss_assemble_f32(const float* nll, const float* kl, const float* kl_weight_dev, const float* g_loss,
                         float* loss, float* rec, float* rowscale, float* dkl, float* acc_dev, int T, int B, void* stream);
/* the same + rng_state[1] += rng_inc (the {seed, offset} state of lv_rng_noise_step, which the fused step then calls with inc = 0) */
int lv_loss_assemble_rng_f32(const float* nll, const float* kl, const float* kl_weight_dev, const float* g_loss,
                             float* loss, float* rec, float* rowscale, float* dkl, float* acc, int T, int B,
                             uint64_t* rng_state, uint64_t rng_inc, void* stream);

/* ---- the batch-sized ends of the two LSTM networks, one launch each (lv_head.hip) --------------------------------------
 * LSTMEncoder's head + GaussianEncoderBase.encode (enc_lstm.py:62-64, encoder.py:40-57): mulv = hT . W_lin^T,
 * z = mu + eps*exp(logvar/2), kl = 0.5*sum(mu^2 + exp(logvar) - logvar - 1).  hT [B][H], W_lin [2nz][H], eps/z [B][ns][nz] */
int lv_enc_head_fwd_f32(const float* hT, const float* w_lin, const float* eps, float* mulv, float* z, float* kl,
                        int B, int H, int ns, int nz, void* stream);
/* ... and its backward: (dz, dkl) -> dmulv [B][2nz], dhT [B][H], gW_lin [2nz][H] ('=').  dz: [dz_parts][B][ns][nz],
 * summed over the leading index in order (1 for a plain gradient; lv_dec_tail_parts(H) after lv_dec_tail_bwd_f32) */
int lv_enc_head_bwd_f32(const float* mulv, const float* eps, const float* dz, int dz_parts, const float* dkl,
                        const float* hT, const float* w_lin, float* dmulv, float* dhT, float* gw_lin, int B, int H, int ns,
                        int nz, void* stream);
/* LSTMDecoder.decode's z-dependent prologue (dec_lstm.py:95-101): c0 = z W_trans^T, h0 = tanh(c0) (G7) and
 * Zp = z W_ih[:, col0:col0+nz]^T + b_ih + b_hh, the contribution of cat((word_embed, z_), -1) to the input projection;
 * Zp gate-major [B][4H], or with column 4u+g (unit_major != 0) for the unit-major Gx epilogue */
int lv_dec_init_f32(const float* z, const float* w_trans, const float* w_ih, long ld_wih, int col0, const float* b_ih,
                    const float* b_hh, float* c0, float* h0, float* zp, int unit_major, int B, int H, int nz, void* stream);
/* ... and its backward from the BPTT's sums: dGsum [B][4H] (gate-major), dc0 [B][H] -> gW_ih[:, col0:col0+nz] (ld_gwih),
 * g_b_ih, g_b_hh, gW_trans ('=') and dz = dGsum . W_ih[:, col0:] + dc0 . W_trans as lv_dec_tail_parts(H) partial sums
 * dz_parts [parts][B][nz] over slices of the contraction (summed in order by lv_enc_head_bwd_f32 / lv_colsum_f32) */
int lv_dec_tail_parts(int H);
int lv_dec_tail_bwd_f32(const float* dGsum, const float* dc0, const float* z, const float* w_ih, long ld_wih, int col0,
                        const float* w_trans, float* gw_ih, long ld_gwih, float* gw_trans, float* gb_ih, float* gb_hh,
                        float* dz_parts, int B, int H, int nz, void* stream);

/* small elementwise / reductions used by the sequencing (h0 = tanh(c0) dec_lstm.py:100; bias grads) */
int lv_tanh_f32(const float* in, float* out, long n, void* stream);
int lv_colsum_f32(const float* in, long ld, int R, int C, float* out, float* out2, void* stream);
int lv_add_f32(const float* a, const float* b, float* out, long n, void* stream);
int lv_sum_accum_f32(const float* x, long n, float* out_dev, void* stream);   /* out_dev[0] += sum(x) (text.py:381) */
int lv_add_scalar_f32(float* x_dev, float v, void* stream);

/* ---- clip_grad_norm_ (text.py:385, image.py:312) + optim.SGD (text.py:325,387) / optim.Adam (image.py:267,314)
 * over flat parameter/gradient buffers */
int lv_sumsq_workspace_floats(void);
int lv_sumsq_f32(const float* x, long n, float* ws, float* out_dev, int accumulate, void* stream);
int lv_clip_coef_f32(const float* sumsq_dev, float max_norm, float* coef_dev, float* norm_out_dev, void* stream);
/* clip_grad_norm_ over two flat gradient buffers at once (encoder + decoder: the norm spans both, G1): one streaming pass
 * over both, then sum + norm + coefficient in one single-workgroup launch.  ws: lv_sumsq_workspace_floats() floats */
int lv_clip_norm2_f32(const float* g1, long n1, const float* g2, long n2, float* ws, float max_norm,
                      float* sumsq_dev, float* coef_dev, float* norm_dev, void* stream);
int lv_sgd_step_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev, int write_back_clipped,
                    void* stream);
int lv_scale_f32(float* x, long n, const float* coef_dev, void* stream);
/* The same clip + update with a device-side TRANSACTION GATE (round 4).  The fused driver keeps the reference's per-iteration
 * host read (text.py:381 loss.sum().item()) off the step, so nothing on the host looks at a step before its update is applied
 * (text.py:385-387); a persistent LSTM launch that hit its bounded hand-off spin must therefore be caught on the DEVICE: the gate
 * runs in the clip coefficient's single thread, reads the engines' status words (and, data parallel, a guard element of the
 * exchanged gradient buffer that lv_txn_guard_f32 set), and either commits the step -- txn[1] += 1, acc[0..2] += the pending
 * (loss, rec, kl) sums txn[2..4] that lv_loss_assemble_f32 left -- or raises the sticky void flag txn[0], under which
 * lv_sgd_step_txn_f32 / lv_scale_txn_f32 are no-ops.  txn: device float[5]; acc: device float[3] or NULL; status / guard may be NULL. */
int lv_clip_norm2_txn_f32(const float* g1, long n1, const float* g2, long n2, float* ws, float max_norm,
                          float* sumsq_dev, float* coef_dev, float* norm_dev, const int* status1, const int* status2,
                          const float* guard, float* txn, float* acc, void* stream);
/* lv_clip_norm2_txn_f32 where part of the gradient arrives as the partial sums of squares its producers emitted (extra[0 .. n_extra):
 * lv_gemm_b16_sumsq, lv_embed_scatter_full_sumsq_f32): g1 / g2 are then the REST of the two flat gradients. */
int lv_clip_norm2_fold_txn_f32(const float* g1, long n1, const float* g2, long n2, float* ws, const float* extra, int n_extra,
                               float max_norm, float* sumsq_dev, float* coef_dev, float* norm_dev, const int* status1,
                               const int* status2, const float* guard, float* txn, float* acc, void* stream);
int lv_clip_coef_txn_f32(const float* sumsq_dev, float max_norm, float* coef_dev, float* norm_out_dev, const int* status1,
                         const int* status2, const float* guard, float* txn, float* acc, void* stream);
int lv_txn_guard_f32(const int* status1, const int* status2, float* guard, void* stream);
int lv_sgd_step_txn_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev, int write_back_clipped,
                        const float* void_flag_dev, void* stream);
int lv_scale_txn_f32(float* x, long n, const float* coef_dev, const float* void_flag_dev, void* stream);
/* lv_sgd_step_txn_f32 on (p, g) and lv_scale_txn_f32 on x2 -- the gradient buffer of the side that is NOT stepped, which
 * clip_grad_norm_ (text.py:385) scales all the same -- in one launch */
int lv_sgd_step_scale_txn_f32(float* p, float* g, long n, const float* lr_dev, const float* coef_dev, int write_back_clipped,
                              float* x2, long n2, const float* void_flag_dev, void* stream);
int lv_adam_step_f32(float* p, float* g, float* m, float* v, long n, const float* lr_dev, const float* coef_dev,
                     const float* step_dev, float beta1, float beta2, float eps, int write_back_clipped, void* stream);

/* ---- Philox4x32-10 draws for throughput mode (stand-in for torch's generator at encoder.py:77, dec_lstm.py:81,106)
 * state_dev: uint64[2] = {seed, offset} in device memory */
int lv_rng_normal_f32(float* out, long n, const uint64_t* state_dev, uint64_t substream, void* stream);
int lv_rng_keepmask_u8(uint8_t* out, long n, float keep_prob, const uint64_t* state_dev, uint64_t substream, void* stream);
int lv_rng_advance(uint64_t* state_dev, uint64_t inc, void* stream);
/* all the noise of one VAE.loss call in one launch: eps (substream 0), dropout_in / dropout_out keep-masks (substreams
 * 1, 2; either may be NULL), then offset += inc (lv_rng_advance queued behind it).  state_dev: uint64[2] = {seed, offset} */
int lv_rng_noise_step(float* eps, long n_eps, uint8_t* mask_in, long n_in, float keep_in, uint8_t* mask_out, long n_out,
                      float keep_out, uint64_t* state_dev, uint64_t inc, void* stream);
int lv_rng_bernoulli_f32(const float* p, float* out, long n, const uint64_t* state_dev, uint64_t substream,
                         void* stream);   /* torch.bernoulli(batch): dynamic binarisation, image.py:287,318 */

/* ---- evaluation statistics on top of the hot path's forward (lv_eval.hip) ----------------------------
 * log q(z|x) of GaussianEncoderBase.eval_inference_dist (modules/encoders/encoder.py:81-109) for z [B][ns][nz] -> out [B][ns];
 * mu == NULL: the standard normal prior of VAE.eval_prior_dist (modules/vae.py:135-145) */
int lv_gauss_logpdf_f32(const float* z, const float* mu, const float* logvar, float* out, int B, int ns, int nz, void* stream);
/* log_sum_exp(value, dim=-1) (modules/utils.py:3-16) over the rows of in [R][C] (ld), plus the constant `add`
 * (VAE.nll_iw, modules/vae.py:127: - log(nsamples)) */
int lv_logsumexp_rows_f32(const float* in, long ld, int R, int C, float add, float* out, void* stream);
/* GaussianEncoderBase.calc_mi (encoder.py:111-145) from the encoder's (mu, logvar) [Bx][nz] and the samples z [Bz][nz]:
 * out_dev[0] = MI estimate, [1] = E log q(z|x), [2] = E log q(z); ws: Bz floats */
int lv_calc_mi_f32(const float* mu, const float* logvar, const float* z, float* ws, float* out_dev, int Bx, int Bz, int nz,
                   void* stream);
/* calc_au (text.py:200-227) accumulators over posterior means mu [B][nz]: acc_dev[k] += sum_b mu[b][k] (mean == NULL) or
 * sum_b (mu[b][k] - mean[k])^2 */
int lv_au_accum_f32(const float* mu, const float* mean, float* acc_dev, int B, int nz, void* stream);

/* ---- generation helpers (lv_eval.hip; SURVEY.md 8f row 4: modules/decoders/dec_lstm.py:163-367) -------------------------
 * torch.argmax(logits, dim=1) (greedy_decode, dec_lstm.py:304): lowest index among equal maxima */
int lv_argmax_rows_f32(const float* in, long ld, int R, int C, int64_t* idx, void* stream);
/* F.log_softmax(logits, -1) + the live hypotheses' running log-probabilities addrow[r] (beam_search_decode, dec_lstm.py:214-218) */
int lv_log_softmax_rows_f32(const float* in, long ld, int R, int C, const float* addrow, float* out, long ldo, void* stream);
/* categorical draw from softmax(logits[r]) by inverse CDF with the uniform u[r] (sample_decode's torch.multinomial,
 * dec_lstm.py:351-352): idx[r] = first c with cumsum softmax >= u[r] */
int lv_sample_rows_f32(const float* in, long ld, int R, int C, const float* u, int64_t* idx, void* stream);

/* ---- Omniglot path: ResNetEncoderV2 (modules/encoders/enc_resnet_v2.py:27-126) and PixelCNNDecoderV2
 * (modules/decoders/dec_pixelcnn_v2.py:12-195).  Activations NHWC ([N*H*W][C]); a convolution = lv_im2col_f32 +
 * lv_gemm_* against weights packed [Cout][taps][Cin]; 1x1 convolutions are plain GEMMs.  `ntaps` = length of the
 * raster-order tap PREFIX that is materialised / used: kh*kw for an ordinary conv, (kh/2)*kw + kw/2 + 1 for a type-B
 * masked conv (dec_pixelcnn_v2.py:17-20) -- the masked taps are skipped, not multiplied by zero. */
int lv_im2col_f32(const float* x, float* col, long ldcol, int N, int H, int W, int C, int Ho, int Wo,
                  int kh, int kw, int pad, int stride, int ntaps, void* stream);
int lv_col2im_f32(const float* dcol, long ldcol, float* dx, int N, int H, int W, int C, int Ho, int Wo,
                  int kh, int kw, int pad, int stride, int ntaps, int accumulate, void* stream);
int lv_conv_pack_w_f32(const float* w /*[Cout][Cin][KK]*/, float* wg /*[Cout][KK][Cin]*/, int Cout, int Cin, int KK, void* stream);
int lv_conv_unpack_dw_f32(const float* dwg, float* dw, int Cout, int Cin, int KK, int accumulate, void* stream);
int lv_mul_inplace_f32(float* w, const float* m, long n, void* stream);   /* MaskedConv2d.forward: weight.data.mul_(mask) */
/* Direct (implicit-GEMM) 32 -> 32 convolutions on 28 x 28 maps: the masked k x k convolutions of PixelCNNBlock
 * (dec_pixelcnn_v2.py:12-62), no im2col buffer; masked taps are skipped in forward / data gradient (the type-B taps are a raster
 * prefix), the weight gradient covers all k*k taps (lv_conv_direct.hip).  in / out NHWC [N][28][28][32]; k odd <= 7. */
long lv_conv32_wpack_floats(int ntaps);
int lv_conv32_pack_f32(const float* w /*[32][32][k*k]*/, float* wp, int k, int ntaps, int transpose, void* stream);
int lv_conv32_f32(const float* in, const float* wp, float* out, int N, int k, int ntaps, int mirror, int accumulate, void* stream);
int lv_conv32_wgrad_slabs(int N, int k);
long lv_conv32_wgrad_ws_floats(int N, int k);
int lv_conv32_wgrad_f32(const float* x, const float* dy, float* dw /*[32][32][k*k]*/, float* ws, int N, int k, int accumulate,
                        void* stream);
/* Pointwise (1 x 1) convolutions between 32 / 64 channels (PixelCNNBlock's bottleneck and expansion convolutions,
 * dec_pixelcnn_v2.py:41-43,49-51): out [P][Cout] (=|+=) in [P][Cin] . W^T, W [Cout][Cin]; w_transposed = 1: W given as
 * [Cin][Cout] (the data gradient of the convolution that owns it); the weight gradient sums dy^T x over all P pixels. */
int lv_conv1x1_f32(const float* in, const float* w, float* out, long P, int Cin, int Cout, int w_transposed, int accumulate,
                   void* stream);
long lv_conv1x1_wgrad_ws_floats(int Cin, int Cout);
/* Split-bf16 forms of lv_conv32_f32 / lv_conv32_bnstat_f32 (round 4): every operand as hi + lo with hi = bf16(x), lo = bf16(x - hi),
 * products on the bf16 matrix pipe with f32 accumulation.  terms = 3: hi*hi' + hi*lo' + lo*hi' (results agree with the exact-f32 form
 * to ~1e-5 relative per product term before summation); terms = 1: plain bf16 operands.  wp16: lv_conv32_pack_b16's image (a buffer of
 * lv_conv32_wpack_floats(ntaps) floats).  bn_partial: NULL, or the BatchNorm stage-1 partials as lv_conv32_bnstat_f32 (forward only). */
int lv_conv32_pack_b16(const float* w, void* wp16, int k, int ntaps, int transpose, void* stream);
int lv_conv32_b16(const float* in, const void* wp16, float* out, float* bn_partial, int N, int k, int ntaps, int mirror, int accumulate,
                  int terms, void* stream);
/* Stage 1 of a BatchNorm BACKWARD in the epilogue of the data-gradient convolution that produces its incoming gradient (round 4;
 * dec_pixelcnn_v2.py:44-62: conv -> BatchNorm -> ELU chains, backward): the data gradient g = dL/d(BN output) leaves the kernel as
 * dv = g * ELU'(y) and every workgroup writes the per-channel (sum dv, sum dv * xhat), xhat = (x - mean) * invstd, of what it stored to
 * partial [blocks][2][C] (blocks = lv_conv32_blocks(N) / lv_conv1x1_blocks(P)); lv_bn_bwd_apply_partials_f32 then finishes the
 * BatchNorm (dx, dgamma, dbeta) with no reduction launch.  y = the BatchNorm's saved output, x = its input.  lv_conv32_bnbwd: terms = 0
 * (exact f32, wp from lv_conv32_pack_f32(transpose = 1)), 1 / 3 (split-bf16, lv_conv32_pack_b16). */
int lv_conv32_bnbwd(const float* in, const void* wp, float* dv, float* partial, int N, int k, int ntaps, const float* y, const float* x,
                    const float* mean, const float* invstd, int act_elu, int terms, void* stream);
int lv_conv1x1_bnbwd_f32(const float* in, const float* w, float* dv, float* partial, long P, int Cin, int Cout, const float* y,
                         const float* x, const float* mean, const float* invstd, int act_elu, void* stream);
int lv_bn_bwd_apply_partials_f32(const float* x, const float* dv, const float* partial, int nblk, const float* mean, const float* invstd,
                                 const float* gamma, float* dx, float* dgamma, float* dbeta, int accumulate_param_grads, long P, int C,
                                 void* stream);
/* ... and of lv_conv32_wgrad_f32 (same scratch and partial layout: lv_wgrad_reduce_batched serves both). */
int lv_conv32_wgrad_b16(const float* x, const float* dy, float* dw, float* ws, int N, int k, int accumulate, int terms, void* stream);
/* All weight-gradient reductions of a backward pass in one launch: lv_conv32_wgrad_f32 / lv_conv1x1_wgrad_f32 called with
 * dw = NULL leave their partial blocks in ws (lv_conv32_wgrad_parts / lv_conv1x1_wgrad_parts of them), and
 * lv_wgrad_reduce_batched sums every layer's partials into its gradient.  desc (HOST memory): 4 int64 per layer =
 * {ws pointer, dw pointer, outputs | parts << 32, k*k (32 -> 32 convolution) or 0 (pointwise)}. */
int lv_conv32_wgrad_parts(int N, int k);
int lv_conv1x1_wgrad_parts(long P);
int lv_wgrad_reduce_batched(const long long* desc, int ndesc, void* stream);
/* PixelCNN ancestral sampling one pixel at a time (dec_pixelcnn_v2.py:201-232; lv_pixelcnn_sample.hip): every layer of
 * PixelCNNDecoderV2.forward at ONE position (i, j) for all B images, from the cached input maps of the masked convolutions at the
 * earlier positions -- bit-equal to the eval-mode full forward (same fma chains in the same order as lv_gemm_f32 /
 * lv_conv1x1_f32 / lv_conv32_f32 / lv_bn_eval_f32).  net: HOST array of lv_pixelcnn_net_words() 64-bit words (device pointers,
 * int64 / double fields of the network table; the per-block tables of lv_pixelcnn_block_words() words each live in device
 * memory); image_engine.PixelCNNSampler builds both.  lv_conv32_tap_split(N): the summation form lv_conv32_f32 uses at batch N. */
int lv_pixelcnn_net_words(void);
int lv_pixelcnn_block_words(void);
int lv_pixelcnn_pixel_step_f32(const long long* net, int B, int i, int j, void* stream);
int lv_conv32_tap_split(int N);
/* Forward convolutions that also leave the stage-1 partials (per-workgroup per-channel sum and sum of squares of their
 * outputs, [blocks][2][Cout]) of the nn.BatchNorm2d that follows them in PixelCNNBlock (dec_pixelcnn_v2.py:41-52), consumed
 * by lv_bn_fwd_partials_f32: the normalisation's statistics pass over the activation is saved. */
int lv_conv32_blocks(int N);
int lv_conv32_bnstat_f32(const float* in, const float* wp, float* out, float* bn_partial, int N, int k, int ntaps, void* stream);
long lv_conv1x1_blocks(long P);
int lv_conv1x1_bnstat_f32(const float* in, const float* w, float* out, float* bn_partial, long P, int Cin, int Cout, void* stream);
int lv_bn_fwd_partials_f32(const float* x, const float* gamma, const float* beta, const float* res, int act_elu, float* y,
                           float* mean, float* invstd, float* run_mean, float* run_var, float eps, float momentum,
                           const float* partial, int nblk, long P, int C, void* stream);
int lv_conv1x1_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, long P, int Cin, int Cout, int accumulate,
                         void* stream);
/* nn.BatchNorm2d in train mode (batch statistics, running stats momentum update with unbiased variance) fused with the
 * residual add and nn.ELU that follow it in ResNetBlock / PixelCNNBlock; backward with ELU' from the saved output */
int lv_bn_workspace_floats(int C);
/* the same module in eval mode (running statistics; image.py:96-187 evaluation passes, dec_pixelcnn_v2.py:201-232 sampling):
 * y = act((x - run_mean) / sqrt(run_var + eps) * gamma + beta (+ res)); mean_out / invstd_out (may be NULL) = what was applied */
int lv_bn_eval_f32(const float* x, const float* gamma, const float* beta, const float* run_mean, const float* run_var, float eps,
                   const float* res, int act_elu, float* y, float* mean_out, float* invstd_out, long P, int C, void* stream);
int lv_bn_fwd_f32(const float* x, const float* gamma, const float* beta, const float* res, int act_elu,
                  float* y, float* mean, float* invstd, float* run_mean, float* run_var,
                  float eps, float momentum, float* ws, long P, int C, void* stream);
int lv_bn_bwd_f32(const float* x, const float* dy, const float* y, const float* mean, const float* invstd,
                  const float* gamma, int act_elu, float* dv, float* dx, float* dgamma, float* dbeta,
                  int accumulate_param_grads, float* ws /* lv_bn_workspace_floats(C) + 2C */, long P, int C, void* stream);
/* The same backward with the incoming gradient given as TWO summands, dy + dy2 (dy2 may be NULL): an activation with two
 * consumers (every residual-block output of dec_pixelcnn_v2.py:33-62) receives two gradient tensors, and adding them while the
 * reduction pass reads them saves the separate accumulation launch (3 passes over the tensor) for one more read. */
int lv_bn_bwd2_f32(const float* x, const float* dy, const float* dy2, const float* y, const float* mean, const float* invstd,
                   const float* gamma, int act_elu, float* dv, float* dx, float* dgamma, float* dbeta,
                   int accumulate_param_grads, float* ws, long P, int C, void* stream);
/* ... and as up to FOUR summands dy + dy2 + dy3 + dy4 (given in order, the unused ones NULL; added left to right): the outputs of
 * the PixelCNN's main blocks feed a residual add, a direct connection and the next block (dec_pixelcnn_v2.py:88-110). */
int lv_bn_bwd4_f32(const float* x, const float* dy, const float* dy2, const float* dy3, const float* dy4, const float* y,
                   const float* mean, const float* invstd, const float* gamma, int act_elu, float* dv, float* dx, float* dgamma,
                   float* dbeta, int accumulate_param_grads, float* ws, long P, int C, void* stream);
/* nn.Sigmoid + the BCE of PixelCNNDecoderV2.reconstruct_error (dec_pixelcnn_v2.py:190-195, eps = 1e-12) */
int lv_sigmoid_bce_fwd_f32(const float* logit, const float* x, float* rec, int B, int npix, float eps, void* stream);
int lv_sigmoid_bce_bwd_f32(const float* logit, const float* x, const float* drec, float* dlogit, int B, int npix,
                           float eps, void* stream);
/* torch.cat([img, z_transform(z).view(B, fm, 28, 28)], dim) as one NHWC tensor (dec_pixelcnn_v2.py:178-186) */
int lv_dec_input_fwd_f32(const float* x, const float* zt, float* in5, int B, int npix, int fm, void* stream);
int lv_dec_input_bwd_f32(const float* din5, float* dzt, int B, int npix, int fm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LVAE_H */
