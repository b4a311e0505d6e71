/* svla.h -- C ABI of the MI355X (gfx950) PPO-Lagrangian hot path for SafeVLA's actor-critic.
 *
 * Boundary contract: plain C entry points, device pointers + sizes + a HIP stream
 * (void* = hipStream_t), return 0 on success, a hipError_t (>0) on launch failure, -1 on invalid arguments.
 * No exceptions cross the boundary; the caller owns every buffer; kernels are re-entrant (one stream per call).
 * bf16 tensors are passed as raw uint16 bits.  The reference has NO native/FFI layer of its own (it is pure
 * Python on top of PyTorch ops), so each entry point below cites the reference *Python* code whose arithmetic it
 * replaces (paths relative to /root/reference).  The reference-side binding a maintainer would add is the ctypes
 * stub in INTEGRATION.md (shipped here as safevla_amd/_lib.py).
 */
#ifndef SVLA_H
#define SVLA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t svla_bf16;

/* Train-mode dropout.  The reference leaves the policy in train() mode (allenact_dino_transformer.py:193), so the dropout 0.1 of
 * nn.TransformerEncoderLayer (attention probabilities, both sub-layer outputs, the feed-forward activation; :545-552) is active in
 * rollouts and updates.  Here it is counter-based and stateless (forward and backward regenerate the same mask, nothing is stored):
 *   r    = mix24((uint32)(e>>1) * 0x9E3779B1 ^ (uint32)(e>>33) * 0x85EBCA77 ^ seed ^ stream * 0xC2B2AE3D)
 *   keep = ((e & 1) ? r >> 16 : r & 0xffff) >= (uint32)(p * 65536 + 0.5);   kept values are scaled by 1 / (1 - p)
 * with e = the element's flat index in its site's logical tensor: (row * row_mult) * N + col for [rows, N] activations (row_mult
 * > 1 when only every row_mult-th row of the logical tensor is materialised), ((row*H + head)*S + query)*S4 + key for attention
 * probabilities with the key stride S rounded up to a multiple of 4.  seed_dev != NULL: the pass seed is read from device memory
 * when the kernel starts instead of `seed` (a captured HIP graph of the launch-bound acting step replays with fresh noise).  mix24(x): x ^= x>>16; x = (x & 0xffffff) * 0xEB352D; x ^= x>>13; x = (x & 0xffffff) * 0x6CA68B; x ^= x>>16 (mod 2^32: the 24-bit multiplies are full-rate on CDNA, 32-bit ones quarter-rate).  NULL / p == 0: no dropout. */
typedef struct svla_dropout { unsigned seed, stream; float p; int row_mult; const unsigned* seed_dev; } svla_dropout;

/* ---- rollout statistics ------------------------------------------------------------------------------ */
/* Reward + cost GAE reverse scan.  Replaces AllenAct-fork RolloutStorage.compute_returns(use_gae=True) [3P];
 * hyper-parameters at training/online/dinov2_vits_tsfm_base.py:345-347; arrays are [T,B] (masks [T+1,B]). */
int svla_gae_scan_f32(const float* rewards, const float* costs, const float* values, const float* c_values,
                      const float* masks, const float* next_v, const float* next_cv, double gamma, double tau, int T, int B,
                      float* ret, float* adv, float* c_ret, float* c_adv, void* stream);

/* ---- losses ---------------------------------------------------------------------------------------------- */
/* SafePPOLogGrad.loss_per_step + loss, forward and backward fused
 * (training/online/loss/customized_loss.py:317-449).  sums[0..2] += {sum (ret-v)^2 (or clipped max), sum action_loss,
 * sum -entropy}; dlogits/dvalues = d total / d(logits, values) with means taken as sum * inv_n. */
int svla_ppo_lag_loss_fwd_bwd_f32(const float* logits, const float* values, const int64_t* actions, const float* old_logp,
                                  const float* adv, const float* c_adv, const float* returns, const float* old_values,
                                  int rows, int A, float lam, float clip, float value_coef, float action_w, float ent_coef,
                                  int use_clipped_value, float inv_n, float* dlogits, float* dvalues, double* sums,
                                  void* stream);
/* PPOValue / SafePPOValue [3P AllenAct fork]; call sites training/online/dinov2_vits_tsfm_base.py:337-342.  old_values != NULL:
 * the clipped form (use_clipped_value_loss=True; the expression of customized_loss.py:374-380 with clip_param = clip). */
int svla_value_mse_fwd_bwd_f32(const float* values, const float* returns, const float* old_values, float clip, int rows, float coef,
                               float inv_n, float* dvalues, double* sums, void* stream);
/* HL-Gauss discrete critic: HLGaussLoss.forward / transform_to_probs / transform_from_probs (utils/loss_functions.py:7-30) and the
 * value read-out of DiscreteCriticHead.forward (architecture/models/allenact_transformer_models/allenact_dino_transformer.py:743-766),
 * forward + backward fused.  logits [rows, NB] (NB <= 256); values_out[r] = sum_j softmax_j * bin_centre_j (optional);
 * target != NULL: sums[0] += cross-entropy(logits[r], hl_gauss_probs(target[r])), dlogits = coef * inv_n * d/dlogits of it;
 * dvalue != NULL: dlogits += dvalue[r] * d values_out[r] / d logits (an MSE-type loss on the read-out value). */
int svla_hlgauss_fwd_bwd_f32(const float* logits, const float* target, const float* dvalue, int rows, int NB, float vmin, float vmax,
                             float sigma, float coef, float inv_n, float* values_out, float* dlogits, double* sums, void* stream);
/* Imitation-learning action loss: nn.CrossEntropyLoss(ignore_index=-1), mean over non-ignored rows
 * (architecture/models/transformer_models/early_fusion_tsfm_models.py:93,115-117).  sums[0] += sum of row losses;
 * dlogits = (softmax - onehot) / *n_valid (device scalar). */
int svla_ce_loss_fwd_bwd_f32(const float* logits, const long* target, int rows, int A, long ignore_index, const float* n_valid,
                             float* dlogits, double* sums, void* stream);

/* ---- heads ------------------------------------------------------------------------------------------------ */
/* LinearActorHead / LinearCriticHead [3P AllenAct] applied at
 * architecture/models/allenact_transformer_models/allenact_dino_transformer.py:441-474.  D <= 1024 (a multiple of 8; 512 for the policy), N <= 20.
 * T > 0: x rows are (b*T + t) (decoder layout), out rows are (t*B + b). */
int svla_small_linear_fwd_f32(const float* x, const float* W, const float* bias, int rows, int N, int D, int T, int B,
                              float* out, void* stream);
int svla_small_linear_bwd_f32(const float* x, const float* W, const float* dout, int rows, int N, int D, int T, int B,
                              int accumulate_dx, float* dx, float* dW, float* db, void* stream);

/* ---- normalisation ---------------------------------------------------------------------------------------- */
/* LayerNorm (rms=0) / RMSNorm (rms=1) rows of width D.  Forward widths 384 / 512 / 768 / 1024 (frozen ViT-S / policy / ViT-B + SigLIP-B + the 768-wide IL presets / ViT-L), backward 512 / 768.  Row maps (G,GS,OFF): logical row m is
 * memory row (m/G)*GS + OFF + m%G (G = 0: identity).  Optional fused ReLU and per-group token add: the
 * "Linear -> LayerNorm -> ReLU (+ camera token)" adapters (allenact_dino_transformer.py:509-513,539-543,672-688);
 * nn.TransformerEncoderLayer norm1/norm2 (:545-552); llama RMSNorm (training/online/third_party_models/llama/model.py:28-71). */
int svla_norm_fwd_bf16(const svla_bf16* x, int xG, int xGS, int xOFF, const float* gamma, const float* beta, float eps, int rows,
                       int D, int rms, int relu, const float* tok, int tok_group, svla_bf16* y, int yG, int yGS, int yOFF,
                       float* mean, float* rstd, void* stream);
int svla_norm_bwd_bf16(const svla_bf16* dy, int dyG, int dyGS, int dyOFF, const svla_bf16* x, int xG, int xGS, int xOFF,
                       const float* gamma, const float* beta, const float* mean, const float* rstd, int rows, int D, int rms,
                       int relu, int tok_group, const svla_bf16* dres, svla_bf16* dx, int dxG, int dxGS, int dxOFF, float* dgamma,
                       float* dbeta, float* dtok, svla_bf16* dx_drop, const svla_dropout* drop, void* stream);

/* ---- GEMMs (MFMA bf16, fp32 accumulate) --------------------------------------------------------------------- */
/* C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T): every nn.Linear / 1x1 nn.Conv2d of the policy
 * (allenact_dino_transformer.py:509-552; llama/model.py:203-222,355-357,437) and, with transposed weights, their input
 * gradients.  act: 0 none, 1 ReLU, 2 GELU(erf).  relu_mask: zero outputs where mask <= 0.  N % 128 == 0, K % 64 == 0.
 * relu_bits_out (act == 1 only): additionally write the sign bits of the output, ceil(M/32)*32 * N/8 bytes in an opaque blocked
 * layout ([M/32][N/64][32 rows][8 bytes], N % 64 == 0; SVLA_RELU_BITS_BYTES); relu_bits: the same mask as relu_mask, given as those bits (the ReLU derivative of nn.TransformerEncoderLayer's
 * feed-forward, allenact_dino_transformer.py:545-552, without re-reading the 2048-wide activation).
 * drop: dropout applied after the activation and before the residual add (sub-layer output / feed-forward activation sites). */
#define SVLA_RELU_BITS_BYTES(M, N) ((((long)(M) + 31) / 32) * 32 * ((long)(N) / 8))
int svla_gemm_nt_bf16(const svla_bf16* A, long lda, const svla_bf16* B, long ldb, const float* bias, const svla_bf16* residual,
                      long ldr, const svla_bf16* relu_mask, long ldm, void* C, long ldc, int M, int N, int K, int act,
                      int out_f32, float alpha, unsigned char* relu_bits_out, const unsigned char* relu_bits,
                      const svla_dropout* drop, void* stream);
/* Kernel choice behind svla_gemm_nt_bf16 (all the same arithmetic; bf16 output, fp32 accumulation, one rounding): generated gfx950 assembly for the row-streaming
 * shapes of the update -- K = 512 without a residual: A-stationary kernels (asmgen/nt_as_gen.py: bias / ReLU + sign bits [+ dropout] / sign-bit mask), and their
 * K = 384 flavours (bias / ReLU + sign bits / erf-GELU: the frozen ViT-S/14's qkv and fc1, the policy's visual compressor); K >= 384 (K % 128 == 0, N % 256 == 0)
 * without dropout: output-stationary kernels (asmgen/nt_os_gen.py: bias and / or residual), from ~4 tiles per CU up.  The A-stationary kernels have two launch
 * shapes: from two 256-row panels per CU up one persistent workgroup per CU sweeps all of N per panel; below that ("mid-M": the 233 panels of the batch-256 probe,
 * the ViT's 216) the grid is (panel slots) x (n-ranges) and a cost model calibrated on profiles/r05_midm_sweep.txt decides between it and the tile kernels (an
 * acting step's 45 panels stay on the tile kernels).  M % 256 tail rows run on the 128-tile kernel -- else the 8-phase 256-tile HIP kernel (>= 160 tiles), else
 * the 128-tile HIP kernel.  ACT_GELU is the erf-GELU of nn.GELU() evaluated as x (1/2 + clamp(t P(z))) with a degree-9 polynomial (asmgen/gelu_poly.py:
 * |error| <= 2.2e-5 for all x, exact tails) in every kernel.
 * Test / A-B hook: on = 1 forces the 128x128-tile kernel, 2 the 256-tile / assembly kernels wherever their shape constraints hold (incl. the mid-M launch whatever the
 * cost model says), 0 = normal dispatch; on = 10 + flags: 8192 = every assembly kernel off, 16384 = the output-stationary ones off (tools/ab_*.py), smaller flags =
 * timing-only ablations of the HIP kernels.  Any value also performs the one-time device-side initialisation of the dispatchers (the binding calls it with 0 at load). */
int svla_gemm_force_small_tile(int on);
/* RMSNorm(A) . W'^T for small M: C[m, n] = epi(rstd[m] * sum_k A[m, k] W'[n, k]), rstd[m] = rsqrt(mean_k A[m, k]^2 + eps), W' = W diag(gamma) folded by the
 * caller -- the pre-norm linears of the frozen T5 encoder block (T5LayerNorm -> q | k | v, -> wi: allenact_dino_transformer.py:506-508 runs HF's T5EncoderModel) and of
 * the llama decoder block (RMSNorm -> wq | wk | wv, -> w1 | w3: training/online/third_party_models/llama/model.py:28-71,170-467) in a KV-cached acting step, where the
 * norm was a launch of its own in front of every such GEMM.  128-tile kernel (the row sums of squares fall out of the A fragments its MFMAs read); epi = +bias[n] ->
 * act -> dropout -> +residual as in svla_gemm_nt_bf16; bf16 output. */
int svla_gemm_nt_rmsa_bf16(const svla_bf16* A, long lda, const svla_bf16* B, long ldb, const float* bias, const svla_bf16* residual, long ldr,
                           void* C, long ldc, int M, int N, int K, int act, float eps, const svla_dropout* drop, void* stream);
/* Test hook: name (NUL-terminated, at most cap - 1 characters) and dispatched M, N, K (mnk[3], may be NULL) of the kernel that the last
 * svla_gemm_nt_bf16 / svla_gemm_tn_f32acc call of this process launched for its main problem ("svla_nt_as_f0", "svla_nt_os_br", "svla_tn_os",
 * "gemm_nt8p_bf16_kernel", "gemm_nt_bf16_kernel", ...): the kernel-choice tests assert the generated assembly really ran (same cited layers). */
int svla_gemm_last_kernel(char* name, int cap, int* mnk);
/* Tool / test hook: launch kernel `name` of the embedded gfx950 assembly code object (safevla_amd/asmgen/) with a raw kernarg block --
 * probes of the assembly builder's instruction semantics on hardware (tools/asm_probe.py); the GEMM entry points above dispatch to the
 * generated kernels themselves (same cited layers). */
int svla_asm_launch_raw(const char* name, const void* kernarg, int kernarg_bytes, int grid, int block, void* stream);
/* dW[N,K] (fp32) += dY[M,N]^T . X[M,K]: weight gradients (autograd of the same layers); optional fused bias gradient
 * db[N] += sum_m dY[m,:] (db may be NULL).  N,K % 128 == 0. */
int svla_gemm_tn_f32acc(const svla_bf16* dY, long ldy, const svla_bf16* X, long ldx, float* dW, long ldw, float* db, int M, int N,
                        int K, void* stream);
/* db[N] += sum_m dY[m*row_stride, :] (bias gradients; row_stride > 1 picks one token of every [S, D] group). */
int svla_colsum_bf16(const svla_bf16* dY, long ldy, int M, int N, int row_stride, float* db, void* stream);

/* fp32 strided GEMM: C[m*ldc + n] (+)= epi(alpha * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk]), epi = +bias[n] -> act (0 none, 1 ReLU,
 * 2 GELU) -> dropout -> zero where mask[m*ldm + n] <= 0 -> +residual[m*ldr + n]; accumulate: C += (weight gradients).  The small fp32
 * heads that are not MFMA-shaped (DiscreteCriticHead / MLPCriticHead, allenact_dino_transformer.py:720-766) and every linear layer of
 * the policy in the fp32 verification mode (same cited layers as svla_gemm_nt_bf16 / svla_gemm_tn_f32acc). */
int svla_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, const float* bias, const float* residual,
                  long ldr, const float* mask, long ldm, float* C, long ldc, int M, int N, int K, int act, int accumulate, float alpha,
                  const svla_dropout* drop, void* stream);
/* out[n] += sum_m X[m*row_stride*ldx + n]: bias gradients of the fp32 path. */
int svla_colsum_f32(const float* X, long ldx, int M, int N, int row_stride, float* out, void* stream);

/* ---- attention ---------------------------------------------------------------------------------------------- */
/* softmax(scale * Q K^T [+ bias] [mask]) V per (row, head), head_dim 64, tokens of one row contiguous (row*S + s).
 * mask_mode 0: none (nn.MultiheadAttention in the fusion encoder, allenact_dino_transformer.py:545-552,702-708);
 * 1: block-causal on traj ids (allenact_dino_transformer.py:398-402 + llama/model.py:317-319).
 * bias [H,S,S] + kvalid [rows,S]: T5 self-attention.  LSE [rows,H,Sq] is saved for the backward.
 * Sq > 0: only the first Sq query tokens of every row are computed (Q/O/dO/dQ hold Sq rows per batch row, Q/dQ row
 * strides ldq/lddq) -- the last fusion layer only feeds sequence position 0 onwards (allenact_dino_transformer.py:708).
 * Sq = 0: all S queries, Q laid out like K/V.  kv_rows > 0 (forward only): K/V hold kv_rows token rows per batch row of
 * which the first S are used -- the llama KV cache of the acting path (llama/model.py:224-239,279-293).
 * Backward: D_ws = optional [rows,H,Sq] fp32 workspace; with it the dQ pass hands rowsum(dO*O) to the dK/dV pass instead of
 * both re-reading O (null: each pass recomputes it). */
int svla_attn_fwd_bf16(const svla_bf16* Q, const svla_bf16* K, const svla_bf16* V, long ld, svla_bf16* O, long ldo, float* LSE,
                       int rows, int S, int H, int head_dim, float scale, int mask_mode, const int* traj, const float* bias,
                       const unsigned char* kvalid, int Sq, long ldq, int kv_rows, const svla_dropout* drop, void* stream);
int svla_attn_bwd_bf16(const svla_bf16* Q, const svla_bf16* K, const svla_bf16* V, long ld, const svla_bf16* O, long ldo,
                       const float* LSE, const svla_bf16* dO, long lddo, svla_bf16* dQ, svla_bf16* dK, svla_bf16* dV, long ldd,
                       int rows, int S, int H, int head_dim, float scale, int mask_mode, const int* traj, const float* bias,
                       const unsigned char* kvalid, int Sq, long ldq, long lddq, float* D_ws, const svla_dropout* drop,
                       void* stream);
/* Test / A-B hook: 1 = the two-kernel backward (dQ kernel + dK/dV kernel, 12 head slices of HBM traffic per (row, head))
 * instead of the single-pass kernel (8 slices) on the unmasked exact-tile shapes; 2 = the single-pass kernel with its (row, head) items in row-major
 * order instead of whole rows per XCD (tools/attn_bwd_once.py: the mapping is worth 1.4 %); 4 = single-query forwards (Sq == 1 without bias / trajectory mask /
 * dropout: the KV-cached acting step) on the tile kernels instead of the decode kernel that reads only the valid keys.  Bits combine. */
int svla_attn_bwd_two_pass(int on);

/* ---- recorded launch sequences --------------------------------------------------------------------------------------------------
 * The single-step acting forward (the reference's rollout collection calls DinoLLAMATxNavActorCritic.forward once per env step,
 * allenact_dino_transformer.py:326-475) and small minibatches are ~100 small dependent kernels per tower: bound by the issue of the
 * launches, not by the kernels.  svla_replay_calls re-issues a recorded sequence of calls to the entry points of THIS header from one
 * C loop (one FFI crossing per sequence instead of one per launch): call i is entry point fn_ids[i] (index in declaration order of
 * this header, this function excluded) with its arguments args[arg_offsets[i] ...] as 64-bit words in declaration order -- pointers and
 * integers by value, float / double by bit pattern.  The dispatcher is generated from this header at build time
 * (safevla_amd/build.py), so the header stays the single source of truth.  Returns 0, or the first non-zero status with
 * *failed_at = the index of the failing call. */
int svla_replay_calls(int n, const int* fn_ids, const int* arg_offsets, const unsigned long long* args, int* failed_at);

/* ---- inputs of a recorded acting step (round 6) ---------------------------------------------------------------------------------------------
 * A recorded single-step forward (svla_replay_calls / svla_replay_calls_grouped) reads its observation from static buffers.  svla_acting_stage fills them in ONE launch
 * from the step's tensors -- the rollout collection of the reference calls the policy once per env step with a fresh observation dict
 * (training/online/allenact_trainer.py via AllenAct's collect_step; forward at allenact_dino_transformer.py:326-475): DINO tokens (tok_bytes of bf16 [B, 2, 84, C]), previous
 * actions [B], masks [B], object-in-hand [B], time step [B], goal token ids [B, L] -> their static copies, the T5 padding mask as int64 and uint8 ((id != 0), column 0 always on),
 * the KV-cache window mask kvalid[b, s] = (s <= t) & (s >= max(t - time_step[b], 0)) for s < max_steps (:388-397), the device-resident step counter t_dev = t, and
 * seed_k += seed_inc (mod 2^32) for the up to three device-resident dropout seeds (NULL: skipped). */
int svla_acting_stage(const void* tok_src, void* tok_dst, long tok_bytes, const int64_t* pa_src, int64_t* pa_dst, const float* mask_src, float* mask_dst,
                      const int64_t* hand_src, int64_t* hand_dst, const int64_t* ts_src, int64_t* ts_dst, const int64_t* ids_src, int64_t* ids_dst, int64_t* am_dst,
                      unsigned char* am8_dst, unsigned char* kvalid_dst, int64_t* t_dev, int B, int L, int max_steps, int t, int* seed0, int* seed1, int* seed2,
                      int seed_inc, void* stream);

/* ---- tower-grouped launches (round 6) ---------------------------------------------------------------------------------------------
 * The three towers of the actor-critic (SafeDinoLLAMATxNavActorCriticSeparate: actor, reward critic, cost critic,
 * architecture/models/allenact_transformer_models/separate_actor_critic.py:27-37) run the same kernel sequence on the same shapes with
 * different weights.  Between svla_group_begin(members) and svla_group_end(stream), the launches of the entry points above are not issued but
 * kept per member (svla_group_member(m) selects whose call comes next); svla_group_end issues launch j of all members as ONE grid -- blockIdx.z /
 * workgroup_id_z picks the member's argument block -- on `stream`, falling back to one launch per member wherever the members' launches differ in
 * anything but their arguments (and for kernels without a grouped twin: csrc/launch.h lists how a kernel takes part).  Results are bit-identical
 * to the per-member launches (same kernels, same arithmetic, same grids per member).  A capture belongs to the calling thread; captures do not nest.
 * svla_replay_calls_grouped replays `members` recorded sequences that name the same entry points in the same order (args[m] = member m's argument
 * words, laid out as for svla_replay_calls) as begin / member calls / end per call index: the acting step of the three towers becomes one dependency
 * chain of grouped launches instead of three streams.  svla_group_stats: launches issued grouped / singly by this thread's captures since the last call. */
int svla_group_begin(int members);
int svla_group_member(int member);
int svla_group_end(void* stream);
int svla_group_stats(long* grouped, long* single);
int svla_replay_calls_grouped(int n, int members, const int* fn_ids, const int* arg_offsets, const unsigned long long* const* args, void* stream,
                              int* failed_at);
/* fn_ids are positions in THIS header: the generated dispatcher reports the hash of the ordered 'name(types)' list it was built from, and the
 * binding (safevla_amd/_lib.py) refuses a library whose stamp differs from the header it parsed. */
int svla_replay_abi_stamp(unsigned long long* stamp);

/* ---- deterministic gradient accumulation --------------------------------------------------------------------------------------
 * Every weight / bias / LayerNorm / embedding gradient of the backward (torch autograd of the layers cited above, e.g.
 * allenact_dino_transformer.py:545-552, feeding the Adam step of training/online/dinov2_vits_tsfm_base.py:331-334) is accumulated across
 * workgroups with fp32 atomics, so its last bits depend on arrival order.  svla_det_config(slot, f32_base, i64_shadow, n) registers
 * an int64 shadow (zero-initialised, n elements) of the fp32 range [f32_base, f32_base + n): from then on every such accumulation
 * whose target lies in a registered range is added to the shadow as 64-bit fixed point (2^-52 resolution; only partials with |partial| < 0.25 enter the shadow, so up to 8192 of them cannot wrap its +-2048 range -- larger or non-finite ones take the plain fp32 atomic and stay visible; integer adds commute,
 * so the sum is bitwise repeatable; each partial is rounded once to the grid, non-finite partials bypass the shadow) instead.
 * svla_det_finalize adds shadow * 2^-52 into the fp32 buffer and clears the shadow.  With it the GRADIENTS are repeatable, and so is the optimiser step: the
 * squared gradient norm behind the clip coefficient is reduced in a fixed order in every mode (svla_sumsq_f32: per-block partials, summed by the last block
 * in slot order).  What still accumulates in arrival order are the loss sums of the info dictionary (fp64 atomics: reported scalars only, no parameter
 * depends on them).
 * slot 0 / 1: two independent ranges (the flat gradient buffer; a scratch range for accumulated intermediates).  NULL, NULL, 0
 * unregisters.  bf16 product path only (the fp32 verification kernels keep their atomics). */
int svla_det_config(int slot, float* f32_base, long long* i64_shadow, long n);
int svla_det_finalize(float* f32, long long* i64_shadow, long n, void* stream);
/* Number of partial sums that had a registered shadow but took the plain fp32 atomic instead (|partial| >= 0.25, NaN, Inf) since the last reset: the run was
 * bitwise repeatable iff 0.  Synchronises the device; the engine reports it as info["det_bypassed_partials"] in deterministic mode. */
int svla_det_bypass_count(unsigned long long* count, int reset);
/* Grid of the shadow: 2^-frac_bits, frac_bits in [36, 52] (default 52); partials with |partial| < 2^(50 - frac_bits) enter it, so 8192 of them cannot wrap the int64.
 * Gradients are 1 / n_total-scaled: the engine lowers frac_bits for small minibatches (52 - ceil(log2(16384 / n_total)), clamped), whose partials are larger -- at 64 rows
 * the default grid sent thousands of partials >= 0.25 around the shadow.  Change it only while the shadows are empty (between updates). */
int svla_det_set_grid(int frac_bits);

/* ---- fp8 attention (BASELINE config 5: "fp8 MFMA attention") ---------------------------------------------------------------
 * The unmasked fusion-encoder attention (same reference op as svla_attn_fwd_bf16: nn.MultiheadAttention inside the post-LN
 * nn.TransformerEncoderLayer, allenact_dino_transformer.py:545-552,702-708) on v_mfma_f32_16x16x32_{fp8,bf8}: Q, K, V and the
 * probabilities in OCP e4m3, dO and dS in OCP e5m2, fp32 accumulation / softmax, bf16 O, dQ, dK, dV.  S <= 256, head_dim 64.
 * SP = S rounded up to 64 / 128 / 192 / 256.
 *   svla_attn_fp8_quant : qkv bf16 [rows*S, ld >= 3*H*64] (Q | K | V column blocks) -> ws (rows*H*6*SP*64 bytes: per (row, head)
 *                         Q8 | Q8T | K8 | K8T | V8 | V8T) and scales [rows*H*3] (dequantisation multiplier of each [S,64] slice)
 *   svla_attn_fp8_fwd   : ws, scales -> O [rows*S, ldo], LSE [rows,H,S] (natural log, like the bf16 kernel)
 *   svla_attn_fp8_bwd   : + O, LSE, dO -> dQ, dK, dV (row stride ldd); gws (rows*H*2*SP*64 bytes), gscale [rows*H] and
 *                         D [rows*H*SP] are scratch the call fills itself (e5m2 copies of dO, their scales, rowsum(dO * O))
 * drop: dropout on the probabilities, the same counter-based masks as the bf16 kernels. */
int svla_attn_fp8_quant(const svla_bf16* qkv, long ld, int rows, int S, int H, int head_dim, unsigned char* ws, float* scales,
                        void* stream);
int svla_attn_fp8_fwd(const unsigned char* ws, const float* scales, svla_bf16* O, long ldo, float* LSE, int rows, int S, int H,
                      int head_dim, float scale, const svla_dropout* drop, void* stream);
int svla_attn_fp8_bwd(const unsigned char* ws, const float* scales, const svla_bf16* O, long ldo, const float* LSE, const svla_bf16* dO,
                      long lddo, unsigned char* gws, float* gscale, float* D, svla_bf16* dQ, svla_bf16* dK, svla_bf16* dV, long ldd,
                      int rows, int S, int H, int head_dim, float scale, const svla_dropout* drop, void* stream);

/* ---- observation / embedding glue ---------------------------------------------------------------------------- */
/* (R,C,7,12) fp32 channels-first DINO features -> bf16 tokens [R, ncam, P, C] (input layout of the 1x1-conv compressor,
 * allenact_dino_transformer.py:663-667; tensor layout per architecture/allenact_preprocessors/dino_preprocessors.py:31-35). */
int svla_feat_to_tokens(const float* feat, int R, int C, int P, int cam, int ncam, svla_bf16* out, void* stream);
/* fusion token + text tokens of the fusion input (allenact_dino_transformer.py:672-692) and the text gradient.  D = the transformer width
 * (512 for the policy and most imitation-learning presets, 768 for the wide ones: early_fusion_tsfm_models.py:275-294), a multiple of 8. */
int svla_fusion_fill(const float* fusion_token, const svla_bf16* text, const int* gid, int R, int S, int L, int text_off, int D,
                     svla_bf16* x0, void* stream);
int svla_fusion_text_bwd(const svla_bf16* dx0, const int* gid, int T, int B, int S, int L, int text_off, int D, float* dtext,
                         void* stream);
/* prev-action (null token where masks == 0) + in-hand embeddings + sinusoidal time encoding
 * (allenact_dino_transformer.py:353-385; architecture/models/transformer_models/text_cond_visual_encoder.py:263-283). */
int svla_decoder_embed_fwd(const svla_bf16* xf, long xf_row_stride, const float* act_tab, const float* hand_tab,
                           const float* div_term, const int64_t* prev_actions, const float* masks, const int64_t* hand,
                           const int64_t* time_step, int T, int B, int n_actions, int D, svla_bf16* out, void* stream);
int svla_decoder_embed_bwd(const svla_bf16* dout, const int64_t* prev_actions, const float* masks, const int64_t* hand, int T,
                           int B, int n_actions, int D, svla_bf16* dxf, long dxf_row_stride, float* d_act_tab, float* d_hand_tab,
                           void* stream);
/* dst[r,:] += src[r,:] on D-wide rows with independent row strides: adds the position-0 gradients of the last fusion layer
 * (allenact_dino_transformer.py:708 keeps only x[:, 0]) into the [R,S,D] input gradient. */
int svla_rows_add_bf16(svla_bf16* dst, long dst_ld, const svla_bf16* src, long src_ld, int rows, int D, void* stream);
/* Zero `bytes` bytes at p on the stream: the scratch the text-gradient scatter (allenact_dino_transformer.py:591-605 backward, svla_fusion_text_bwd)
 * accumulates into -- part of the launch sequence rather than a framework-side fill. */
int svla_zero_bytes(void* p, long bytes, void* stream);
/* llama KV-cache append of the single-step (acting) path, training/online/third_party_models/llama/model.py:279-293
 * (cache_k[:bsz, start_pos] = xk): cache[b, *t_dev, 0:width] = src[b*ld_src + 0:width]; the slot is read from device memory so that a
 * recorded launch sequence of the step does not depend on the step counter. */
int svla_kv_append_bf16(const svla_bf16* src, long ld_src, svla_bf16* cache, long cache_rows, int width, const int64_t* t_dev, int B,
                        void* stream);
/* llama FeedForward gate silu(a)*b (llama/model.py:359-360) on [a | b] rows. */
int svla_swiglu_fwd(const svla_bf16* ab, long M, int Hd, svla_bf16* g, void* stream);
int svla_swiglu_bwd(const svla_bf16* ab, const svla_bf16* dg, long M, int Hd, svla_bf16* dab, void* stream);
/* 64-bit content hash of fixed-width byte rows: de-duplicates the per-row goal strings that the reference tokenises
 * one by one on the host (allenact_dino_transformer.py:591-603; row format utils/string_utils.py:11-18). */
int svla_row_hash_u8(const unsigned char* rows, long n_rows, int row_bytes, int64_t* out, void* stream);
/* T5 shared-embedding gather (HF T5EncoderModel called at allenact_dino_transformer.py:603). */
int svla_embed_gather_f32_bf16(const float* table, const int64_t* ids, long n, int D, svla_bf16* out, void* stream);

/* ---- frozen ViT preprocessor (rollout time) ------------------------------------------------------------------- */
/* DataAugmentationPreprocessor.process without augmentation: (x/255 - mean)/std
 * (architecture/allenact_preprocessors/dino_preprocessors.py:224-239; DINO_RGB_MEANS/STDS :42-43).  x 4-byte aligned, y 16-byte aligned
 * (one dword of the HWC stream per lane -> one float4 per lane). */
int svla_normalize_u8_f32(const unsigned char* x, long n, const float* mean3, const float* std3, float* y, void* stream);
/* The same normalisation fused with the W crop [3:-3] and the 14x14/14 patch-embedding im2col of DinoViTEmbedder.forward
 * (dino_preprocessors.py:27-35): u8 HWC frames -> bf16 rows [B, gh*gw, KP], k = c*P*P + ky*P + kx, zero padded to KP (KP % 8 == 0: every lane
 * writes 16 bytes of an im2col row; the image rows are fetched as aligned dwords). */
int svla_patchify_u8_bf16(const unsigned char* frames, int B, int H, int W, int crop_x, int P, int gh, int gw, int KP,
                          const float* mean3, const float* std3, svla_bf16* out, void* stream);
/* cls token + position embedding (DINOv2 prepare_tokens [3P torch.hub facebookresearch/dinov2], dino_preprocessors.py:106);
 * cls == NULL: no class token -- y[b, p] = patch[b, p] + pos[p] (timm SigLIP trunk [3P], architecture/allenact_preprocessors/siglip_preprocessors.py:86-88). */
int svla_vit_tokens(const svla_bf16* patch, const float* cls, const float* pos, int B, int NP, int C, svla_bf16* y, void* stream);
/* x_norm_patchtokens -> (B,C,16,27) -> AdaptiveAvgPool2d((7,12)) (dino_preprocessors.py:24,31-35); bf16 tokens and/or fp32 CHW. */
int svla_adaptive_pool_tokens(const svla_bf16* x, int B, int skip, int gh, int gw, int C, int oh, int ow, int cam, int ncam,
                              svla_bf16* tok_out, float* chw_out, void* stream);

/* In-place dropout of a [rows, N] bf16 activation, element index row*N + col: the two stand-alone sites of the frozen T5 encoder
 * (after the token embedding, after the final layer norm) that stays in train() mode with the rest of the policy
 * (allenact_dino_transformer.py:193,599-603). */
int svla_dropout_bf16(svla_bf16* x, long rows, int N, const svla_dropout* drop, void* stream);

/* ---- optimiser (Adam lr 2e-5, max_grad_norm 0.5: training/online/dinov2_vits_tsfm_base.py:331-334; weight_decay > 0 = the
 * decoupled AdamW of the imitation-learning trainer, training/offline/train_pl.py:283-287) ---------------------------------- */
/* *out += sum g[i]^2, reduced in a fixed order (block partials in a library-owned scratch, summed by the block that arrives last): calls that share `out`
 * must be issued on one stream (the engine's per-tower calls are).  The scratch is per (device, stream) -- launches in flight on different streams do not
 * share partials or the arrival counter -- and is allocated by the first call on that stream (not inside a stream capture). */
int svla_sumsq_f32(const float* g, long n, double* out, void* stream);
int svla_adam_step_f32(float* p, const float* g, float* m, float* v, svla_bf16* p_bf16, long n, float lr, float beta1,
                       float beta2, float eps, int step, const double* gnorm_sq, float max_norm, float grad_scale,
                       float weight_decay, void* stream);
int svla_cast_f32_bf16(const float* src, svla_bf16* dst, long n, void* stream);
int svla_transpose_cast_f32_bf16(const float* src, int rows, int cols, svla_bf16* dst, void* stream);

/* ---- fp32 verification mode --------------------------------------------------------------------------------------------------
 * north_star asks for "matching losses/entropy within fp32 tolerance"; the reference computes everything in fp32 (no autocast anywhere,
 * SURVEY 8).  The entry points below are fp32-activation twins of the bf16 ones above (same argument meaning, same cited reference
 * layers: norms allenact_dino_transformer.py:509-552 + llama/model.py:28-71; attention allenact_dino_transformer.py:545-552,398-402 +
 * llama/model.py:249-322; glue allenact_dino_transformer.py:353-385,663-692 + llama/model.py:359-360); with svla_gemm_f32 they run the
 * SAME host schedule (model.precision = "fp32") so that logits / values / losses / lambda after an update can be checked against the
 * reference-generated goldens at 1e-4 instead of the bf16 ladder.  Simple kernels, not tuned: a verification path, not the product path. */
int svla_norm_fwd_f32(const float* x, int xG, int xGS, int xOFF, const float* gamma, const float* beta, float eps, int rows, int D,
                      int rms, int relu, const float* tok, int tok_group, float* y, int yG, int yGS, int yOFF, float* mean, float* rstd,
                      void* stream);
int svla_norm_bwd_f32(const float* dy, int dyG, int dyGS, int dyOFF, const float* x, int xG, int xGS, int xOFF, const float* gamma,
                      const float* beta, const float* mean, const float* rstd, int rows, int D, int rms, int relu, int tok_group,
                      const float* dres, float* dx, int dxG, int dxGS, int dxOFF, float* dgamma, float* dbeta, float* dtok, float* dx_drop,
                      const svla_dropout* drop, void* stream);
/* fp32 attention: head_dim <= 128 (64 for the policy; 96 for the imitation-learning presets with TransformerConfig(n, 768, 8), early_fusion_tsfm_models.py:236-240,275-279,
 * whose bf16 activations take these kernels through fp32 copies: the MFMA kernels above are built for 64-wide heads), S <= 512. */
int svla_attn_fwd_f32(const float* Q, const float* K, const float* V, long ld, float* O, long ldo, float* LSE, int rows, int S, int H,
                      int head_dim, float scale, int mask_mode, const int* traj, const float* bias, const unsigned char* kvalid, int Sq,
                      long ldq, int kv_rows, const svla_dropout* drop, void* stream);
int svla_attn_bwd_f32(const float* Q, const float* K, const float* V, long ld, const float* O, long ldo, const float* LSE, const float* dO,
                      long lddo, float* dQ, float* dK, float* dV, long ldd, int rows, int S, int H, int head_dim, float scale, int mask_mode,
                      const int* traj, const unsigned char* kvalid, int Sq, long ldq, long lddq, const svla_dropout* drop, void* stream);
int svla_feat_to_tokens_f32(const float* feat, int R, int C, int P, int cam, int ncam, float* out, void* stream);
int svla_fusion_fill_f32(const float* fusion_token, const float* text, const int* gid, int R, int S, int L, int text_off, int D, float* x0,
                         void* stream);
int svla_fusion_text_bwd_f32(const float* dx0, const int* gid, int T, int B, int S, int L, int text_off, int D, float* dtext, void* stream);
int svla_decoder_embed_fwd_f32(const float* xf, long xf_row_stride, const float* act_tab, const float* hand_tab, const float* div_term,
                               const int64_t* prev_actions, const float* masks, const int64_t* hand, const int64_t* time_step, int T, int B,
                               int n_actions, int D, float* out, void* stream);
int svla_decoder_embed_bwd_f32(const float* dout, const int64_t* prev_actions, const float* masks, const int64_t* hand, int T, int B,
                               int n_actions, int D, float* dxf, long dxf_row_stride, float* d_act_tab, float* d_hand_tab, void* stream);
int svla_rows_add_f32(float* dst, long dst_ld, const float* src, long src_ld, int rows, int D, void* stream);
int svla_swiglu_fwd_f32(const float* ab, long M, int Hd, float* g, void* stream);
int svla_swiglu_bwd_f32(const float* ab, const float* dg, long M, int Hd, float* dab, void* stream);
int svla_embed_gather_f32(const float* table, const int64_t* ids, long n, int D, float* out, void* stream);
int svla_dropout_f32(float* x, long rows, int N, const svla_dropout* drop, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVLA_H */
