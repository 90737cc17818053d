/*
 * rlpyt_hip.h -- C ABI of librlpyt_hip.so: the MI355X (gfx950 / CDNA4) hot path of
 * astooke/rlpyt (rollout -> advantage/return -> minibatch loss; prioritized replay).
 *
 * The reference is pure Python: it has no FFI.  Each entry point below replaces the
 * body of one reference routine (cited as file:line under /root/reference); the
 * reference-side binding a maintainer would add is the ctypes stub shown in
 * INTEGRATION.md (rlpyt_amd/_lib.py is that stub, in tree).
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer (HBM) unless the name ends in
 *     "_host"; arrays are dense row-major with the leading dims stated per call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is
 *     asynchronous with respect to the host unless stated otherwise;
 *   - no torch types, no allocation inside a call except the opaque handles
 *     (`rlpyt_sumtree`) which own their HBM;
 *   - return value: 0 = ok, <0 = RLPYT_E*; `rlpyt_hip_last_error()` gives the text
 *     (thread-local);
 *   - `done`/`valid` masks: `done` is uint8 0/1 (torch.bool storage), `valid` is f32 0/1
 *     exactly as the reference materialises them.
 */
#ifndef RLPYT_HIP_H
#define RLPYT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLPYT_OK 0
#define RLPYT_EINVAL (-1) /* bad argument (null pointer, negative size, unsupported option) */
#define RLPYT_ESHAPE (-2) /* shape / alignment not supported by the requested variant */
#define RLPYT_EHIP (-3)   /* a HIP runtime call failed; see rlpyt_hip_last_error() */
#define RLPYT_ESTATE (-4) /* handle used out of protocol (e.g. update before sample) */
#define RLPYT_ETIMEOUT (-5) /* rlpyt_seq_wait gave up (peer process gone?) */

typedef void* rlpyt_stream_t; /* hipStream_t */

const char* rlpyt_hip_last_error(void);
/* ABI version of this header (3); bumped when a signature, an entry point or a workspace layout
 * changes, so that a stale .so fails the binding's version check instead of an attribute lookup. */
#define RLPYT_HIP_ABI_VERSION 17
int rlpyt_hip_abi_version(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; returns CU count or <0. */
int rlpyt_hip_device_info(char* name, int cap);

/* Which kernel variant ran.  Several entry points pick between size / alignment-gated kernel
 * instantiations (e.g. rlpyt_frames_gather_seq switches to a wide kernel at n*seq_T >= 2048);
 * every launch is counted per instantiation so that a parity test can assert WHICH one produced
 * the result it checked (tests/test_variants.py).  Names are the device kernels' demangled names
 * without namespaces / argument lists, e.g. "scan_exact_kernel<0, 4, 8, true>".
 *   rlpyt_hip_last_variant()   name of the last kernel this thread launched ("" if none);
 *   rlpyt_hip_variant_reset()  zero all counters;
 *   rlpyt_hip_variant_dump()   "name\tcount\n" lines for every instantiation launched since the
 *                              reset; returns the bytes needed (call with buf = NULL to size).
 * A launch recorded into a hipGraph counts once, at capture.  Host-side, no HIP stream work. */
const char* rlpyt_hip_last_variant(void);
void rlpyt_hip_variant_reset(void);
int64_t rlpyt_hip_variant_dump(char* buf, int64_t cap);

/* Page-lock (pin) an existing host range so the sampler's fork-shared step buffer
 * (rlpyt/samplers/parallel/gpu/sampler.py:134-139) can be the source / destination of
 * asynchronous H2D / D2H copies.  Host pointers; synchronous. */
int rlpyt_host_register(void* host_ptr, int64_t bytes);
int rlpyt_host_unregister(void* host_ptr);
/* Device-side address of a range pinned with rlpyt_host_register (mapped into the device's
 * address space): kernels may read the workers' newest frames and write the sampled actions
 * in place -- no staging copy, no DMA descriptor latency on the per-step critical path. */
int rlpyt_host_device_pointer(void* host_ptr, void** dev_ptr);

/* Step hand-off between the sampler master and its forked env workers: replaces the
 * 2 x n_workers semaphores per time step of rlpyt/samplers/parallel/gpu/action_server.py:44-58
 * / collectors.py:29-50 with 32-bit sequence words in fork-shared memory (futex; host only,
 * no HIP call -- usable in forked children).
 *   rlpyt_seq_post(word, v)       master: publish step number v, wake every waiter;
 *   rlpyt_seq_wait(word, v, ...)  block until (int32)(*word - v) >= 0 (spin, then sleep);
 *                                 timeout_ms <= 0 waits forever, else RLPYT_ETIMEOUT;
 *   rlpyt_seq_arrive(word, n)     worker: atomically ++*word, wake the master once it
 *                                 reaches n (= arrivals expected so far). */
int rlpyt_seq_wait(uint32_t* word, uint32_t target, int spin_iters, int timeout_ms);
int rlpyt_seq_post(uint32_t* word, uint32_t value);
int rlpyt_seq_arrive(uint32_t* word, uint32_t wake_at);

/* ------------------------------------------------------------------------------------
 * Return / advantage scans over [T, N] trajectories (N = B * any trailing dims).
 * Coalesced along N, sequential along T.
 * ---------------------------------------------------------------------------------- */

/* Scan variants. */
#define RLPYT_SCAN_EXACT 0     /* one lane per column, reference association, no FMA
                                  contraction: bit-identical to the reference fp32 path */
#define RLPYT_SCAN_SEGMENTED 1 /* time axis split in LDS-staged segments, affine-map
                                  composition across segments (wave shuffles): lower
                                  latency at small N, fp32 tolerance (re-associated) */

/* generalized_advantage_estimation -- rlpyt/algos/utils.py:24-40.
 *   A[T-1] = r + g*bv*nd - V ; A[t] = (r + g*V[t+1]*nd - V) + (g*l)*nd*A[t+1] ; R = A + V
 * `valid` (nullable) additionally receives valid_from_done(done) (utils.py:104-112),
 * fusing rlpyt/algos/pg/base.py:57-63 into one launch. */
int rlpyt_gae_f32(const float* reward, const float* value, const uint8_t* done,
                  const float* bootstrap /*[N]*/, float* advantage, float* return_,
                  float* valid /*nullable [T,N]*/, int T, int64_t N, double discount,
                  double gae_lambda, int variant, rlpyt_stream_t stream);

/* discount_return -- rlpyt/algos/utils.py:8-21.  R[t] = r + R[t+1]*g*nd.
 * If `value` and `advantage` are non-null also writes advantage = R - V
 * (rlpyt/algos/pg/base.py:53-55).  `valid` nullable as above. */
int rlpyt_discount_return_f32(const float* reward, const uint8_t* done,
                              const float* bootstrap /*[N]*/, float* return_,
                              const float* value /*nullable*/, float* advantage /*nullable*/,
                              float* valid /*nullable*/, int T, int64_t N, double discount,
                              int variant, rlpyt_stream_t stream);

/* valid_from_done -- rlpyt/algos/utils.py:104-112. valid[0]=1, valid[t]=1-min(1,sum done[:t]) */
int rlpyt_valid_from_done(const uint8_t* done, float* valid, int T, int64_t N,
                          rlpyt_stream_t stream);

/* discount_return_n_step -- rlpyt/algos/utils.py:67-101.
 * in: reward f32 [T_in,N], done u8 [T_in,N]; out: return_ f32 [T_out,N], done_n u8 [T_out,N]
 * with T_out = do_truncated ? T_in : T_in-(n_step-1). */
int rlpyt_nstep_return_f32(const float* reward, const uint8_t* done, float* return_,
                           uint8_t* done_n, int T_in, int64_t N, int n_step, double discount,
                           int do_truncated, rlpyt_stream_t stream);

/* Advantage normalisation -- rlpyt/algos/pg/base.py:65-73:
 *   A <- (A - mean(A[valid>0])) / max(std_unbiased(A[valid>0]), eps), in place.
 * `workspace`: >= rlpyt_adv_normalize_workspace_bytes(n) bytes of device scratch.
 * `stats_out` (nullable, device, 3 floats): mean, std, count. */
int64_t rlpyt_adv_normalize_workspace_bytes(int64_t n);
int rlpyt_adv_normalize_f32(float* advantage, const float* valid /*nullable*/, int64_t n,
                            float eps, void* workspace, float* stats_out,
                            rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Policy-gradient losses, forward + backward in one pass over the minibatch.
 * out_scalars (device, 5 floats): loss, pi_loss, value_loss, entropy, perplexity.
 * grad_prob [M,A], grad_value [M] receive dLoss/dprob_new and dLoss/dvalue.
 * `workspace`: >= rlpyt_pg_loss_workspace_bytes(M) bytes of device scratch.
 * ---------------------------------------------------------------------------------- */
int64_t rlpyt_pg_loss_workspace_bytes(int64_t M);

/* PPO.loss -- rlpyt/algos/pg/ppo.py:117-154 with Categorical
 * (rlpyt/distributions/categorical.py:32-43, EPS=1e-8) and valid_mean
 * (rlpyt/utils/tensor.py:39-46). */
int rlpyt_ppo_loss_fwd_bwd_f32(const float* prob_new /*[M,A]*/, const float* value /*[M]*/,
                               const float* prob_old /*[M,A]*/, const int64_t* action /*[M]*/,
                               const float* advantage, const float* return_,
                               const float* valid /*nullable [M]*/, int64_t M, int A,
                               float ratio_clip, float value_loss_coeff,
                               float entropy_loss_coeff, float* out_scalars,
                               float* grad_prob, float* grad_value, void* workspace,
                               rlpyt_stream_t stream);

/* PPO loss with the policy / value heads fused in -- rlpyt/models/pg/atari_ff_model.py:56-58 +
 * rlpyt/algos/pg/ppo.py:133-153, forward and backward in one pass over the trunk output:
 *   logits = h w_pi^T + b_pi; pi = softmax(logits); v = h w_v^T + b_v; PPO loss as above.
 * h f32 [M,K] (K = 256 or 512), w_pi [A,K] (A <= 8), w_v [K].  Outputs: out_scalars[5] as
 * rlpyt_ppo_loss_fwd_bwd_f32; grad_h [M,K] = dL/dh; grad_params [A*K + K + A + 1] =
 * dL/dw_pi | dL/dw_v | dL/db_pi | dL/db_v (fixed-order partial reduction: deterministic).
 * With flat_idx [M] the per-sample inputs (prob_old, action, advantage, return_) are the
 * [T,B,...] batch arrays, read at (idx % T, idx / T) inside the kernel (no gather launches);
 * valid must then be NULL.
 * workspace: rlpyt_ppo_head_loss_workspace_bytes(K, A) bytes. */
int64_t rlpyt_ppo_head_loss_workspace_bytes(int K, int A);
int rlpyt_ppo_head_loss_fwd_bwd_f32(const float* h, const float* w_pi, const float* b_pi,
                                    const float* w_v, const float* b_v, const float* prob_old,
                                    const int64_t* action, const float* advantage,
                                    const float* return_, const float* valid /*nullable*/,
                                    const int64_t* flat_idx /*nullable*/, int T, int64_t B,
                                    int64_t M, int K, int A, float ratio_clip,
                                    float value_loss_coeff, float entropy_loss_coeff,
                                    float* out_scalars, float* grad_h, float* grad_params,
                                    void* workspace, rlpyt_stream_t stream);

/* The same with the trunk's bias add + ReLU fused in (rlpyt/models/mlp.py:24-31: the Linear + ReLU
 * pair in front of the heads): ``z`` f32 [M,K] is the trunk's pre-activation WITHOUT its bias
 * (z = x W^T), trunk_bias [K]; the kernel applies h = relu(z + trunk_bias) while loading a row.
 * grad_z [M,K] = dL/dz (= dL/dh masked by h > 0); grad_params [A*K + K + A + 1 + K] =
 * dL/dw_pi | dL/dw_v | dL/db_pi | dL/db_v | dL/dtrunk_bias.  trunk_bias == NULL: exactly
 * rlpyt_ppo_head_loss_fwd_bwd_f32 (z is then the post-activation trunk output, grad_params has no
 * trunk-bias block). */
int rlpyt_ppo_trunk_head_loss_fwd_bwd_f32(const float* z, const float* trunk_bias /*nullable*/,
                                          const float* w_pi, const float* b_pi, const float* w_v,
                                          const float* b_v, const float* prob_old,
                                          const int64_t* action, const float* advantage,
                                          const float* return_, const float* valid /*nullable*/,
                                          const int64_t* flat_idx /*nullable*/, int T, int64_t B,
                                          int64_t M, int K, int A, float ratio_clip,
                                          float value_loss_coeff, float entropy_loss_coeff,
                                          float* out_scalars, float* grad_z, float* grad_params,
                                          void* workspace, rlpyt_stream_t stream);

/* The same for CAPTURED update graphs: ratio_clip_dev (nullable, device, 1 float) overrides
 * ratio_clip at run time, so that one captured launch serves the reference's linear clip schedule
 * (rlpyt/algos/pg/ppo.py:112-114). */
int rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32(
    const float* z, const float* trunk_bias /*nullable*/, const float* w_pi, const float* b_pi,
    const float* w_v, const float* b_v, const float* prob_old, const int64_t* action,
    const float* advantage, const float* return_, const float* valid /*nullable*/,
    const int64_t* flat_idx /*nullable*/, int T, int64_t B, int64_t M, int K, int A,
    float ratio_clip, const float* ratio_clip_dev /*nullable*/, float value_loss_coeff,
    float entropy_loss_coeff, float* out_scalars, float* grad_z, float* grad_params,
    void* workspace, rlpyt_stream_t stream);

/* c[M,N] = a[M,K] * b[N,K]^T, all f32 row-major -- the trunk Linear of the update
 * (rlpyt/models/mlp.py:24-31 via torch.nn.functional.linear: forward x W^T; with b = W^T the input
 * gradient g W).  Computed on the bf16 matrix pipe from exact three-piece bf16 splits of both
 * operands (six products of order <= 2, f32 accumulation; dropped terms <= 2^-24 |ab|, 2^-27 rms): f32 in,
 * f32 out, f32-level error.  K must be a multiple of 32; a, b 16-byte aligned. */
int rlpyt_gemm_nt_f32(const float* a, const float* b, float* c, int64_t M, int64_t N, int64_t K,
                      rlpyt_stream_t stream);
/* The weight gradient of the same Linear (rlpyt/models/mlp.py:24-31 under autograd), same
 * arithmetic: c[M,N] = a[K,M]^T * b[K,N] -- g^T x, a contraction over the batch axis; all f32
 * row-major, K a multiple of 32, M and N multiples of 4, pointers 16-byte aligned.  For K >= 2048
 * K is cut into 8 chunks (one per XCD) whose partial tiles go to `workspace`
 * (rlpyt_gemm_tn_workspace_bytes; 0 = none needed) and are summed in a fixed order: results are
 * run-to-run identical. */
int64_t rlpyt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K);
int rlpyt_gemm_tn_f32(const float* a, const float* b, float* c, int64_t M, int64_t N, int64_t K,
                      void* workspace, rlpyt_stream_t stream);

/* A2C.loss -- rlpyt/algos/pg/a2c.py:63-103: pi_loss = -valid_mean(log(p[a]+eps) * A). */
int rlpyt_a2c_loss_fwd_bwd_f32(const float* prob /*[M,A]*/, const float* value /*[M]*/,
                               const int64_t* action, const float* advantage,
                               const float* return_, const float* valid /*nullable*/,
                               int64_t M, int A, float value_loss_coeff,
                               float entropy_loss_coeff, float* out_scalars, float* grad_prob,
                               float* grad_value, void* workspace, rlpyt_stream_t stream);

/* DQN.loss -- rlpyt/algos/dqn/dqn.py:231-263 (Huber TD with IS weights).
 *   q = qs[i,a_i]; tq = double ? target_qs[i, argmax next_qs[i]] : max target_qs[i]
 *   y = return_ + (1-done_n) * disc_n * tq ; delta = y - q
 *   losses = huber(delta, delta_clip) (delta_clip<=0: 0.5*delta^2) * is_weights ; loss = mean
 * out_scalars (2 floats): loss, mean |delta|.  td_abs [M] = clamp(|delta|, 0, delta_clip).
 * grad_qs [M,A] = dLoss/dqs (zero except the taken action). */
int rlpyt_dqn_loss_fwd_bwd_f32(const float* qs /*[M,A]*/, const float* target_qs /*[M,A]*/,
                               const float* next_qs /*nullable [M,A]: double-DQN*/,
                               const int64_t* action, const float* return_,
                               const uint8_t* done_n, const float* is_weights /*nullable*/,
                               int64_t M, int A, float disc_n, float delta_clip,
                               float* out_scalars, float* td_abs, float* grad_qs,
                               void* workspace, rlpyt_stream_t stream);

/* R2D1.loss -- rlpyt/algos/dqn/r2d1.py:298-345 (after the network forward passes).
 *   tq = double ? target_qs[argmax next_qs] : max target_qs       (all [T,B,A])
 *   y = h(return_ + (1-done_n) * disc_n * h^-1(tq)); delta = y - qs[action]
 *   losses = (delta_clip<=0 ? 0.5 delta^2 : huber) * is_weights[b]; loss = valid_mean
 *   td = clamp(|delta|) * valid ; priorities[b] = eta*max_t td + (1-eta)*valid_mean_t |delta|
 * out_scalars (2 floats): loss, 1/sum(valid).  grad_qs [T,B,A] = dLoss/dqs. */
int64_t rlpyt_r2d1_loss_workspace_bytes(void);
int rlpyt_r2d1_loss_fwd_bwd_f32(const float* qs, const float* target_qs,
                                const float* next_qs /*nullable*/, const int64_t* action,
                                const float* return_, const uint8_t* done_n,
                                const float* valid /*[T,B]*/,
                                const float* is_weights /*nullable [B]*/, int T, int B, int A,
                                float disc_n, float delta_clip, float value_scale_eps,
                                float pri_eta, float* out_scalars, float* td_abs_valid,
                                float* priorities /*[B]*/, float* grad_qs, void* workspace,
                                rlpyt_stream_t stream);

/* CategoricalDQN.loss -- rlpyt/algos/dqn/cat_dqn.py:34-93 (after the network forward passes).
 *   next_z[j]  = clamp(return_ + (1-done_n) * (z[j] * disc_n), V_min, V_max)
 *   a'         = argmax_a sum_j sel[a,j] z[j], sel = next_ps (double DQN) or target_ps
 *   target_p[i]= sum_j target_ps[a',j] * clamp(1 - |next_z[j] - z[i]| / delta_z, 0, 1)
 *   p          = clamp(ps[action], 1e-6, 1);  losses = -sum_i target_p[i] log p[i] (* is_weights)
 *   KL         = clamp(sum_i tc[i] (log tc[i] - log p[i]), 1e-6, 1e6), tc = clamp(target_p, 1e-6, 1)
 *   loss       = mean(losses), or with `valid`: sum(losses*valid)/sum(valid) and KL *= valid
 * ps / target_ps / next_ps f32 [M,A,P] probabilities (P <= 64 atoms), z f32 [P] the atom grid
 * (torch.linspace(V_min, V_max, P)), delta_z = (V_max - V_min)/(P-1).
 * out_scalars (2 floats): loss, normaliser (M or sum(valid)).  kl_div [M].
 * grad_ps [M,A,P] = dLoss/dps (zero except the taken action's row; zero where the 1e-6 / 1
 * clamp of p is active, as torch.clamp's backward). */
int64_t rlpyt_cat_dqn_loss_workspace_bytes(void);
int rlpyt_cat_dqn_loss_fwd_bwd_f32(const float* ps, const float* target_ps,
                                   const float* next_ps /*nullable*/, const int64_t* action,
                                   const float* return_, const uint8_t* done_n,
                                   const float* is_weights /*nullable [M]*/,
                                   const float* valid /*nullable [M]*/, const float* z /*[P]*/,
                                   int64_t M, int A, int P, float v_min, float v_max,
                                   float disc_n, float* out_scalars, float* kl_div,
                                   float* grad_ps, void* workspace, rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Observation running mean / std -- rlpyt/models/running_mean_std.py:21-45 and the
 * normalise+clip of rlpyt/models/pg/mujoco_ff_model.py:68-73.  x is [n, D] row-major.
 * ---------------------------------------------------------------------------------- */
int64_t rlpyt_obs_rms_workspace_bytes(int64_t n, int64_t D);
/* per-dimension mean and BIASED variance over the n rows. */
int rlpyt_obs_batch_stats_f32(const float* x, int64_t n, int64_t D, float* mean, float* var,
                              void* workspace, rlpyt_stream_t stream);
/* Chan merge of (batch_mean, batch_var, batch_count) into running (mean, var, count[1]). */
int rlpyt_obs_rms_merge_f32(float* mean, float* var, float* count, const float* batch_mean,
                            const float* batch_var, float batch_count, int64_t D,
                            rlpyt_stream_t stream);
/* out = clamp((x - mean) / sqrt(max(var, var_clip)), -obs_clip, obs_clip); var_clip<=0: none */
int rlpyt_obs_normalize_f32(const float* x, const float* mean, const float* var, float* out,
                            int64_t n, int64_t D, float var_clip, float obs_clip,
                            rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gathers.
 * ---------------------------------------------------------------------------------- */

/* Minibatch row gather out of a [T,B,E] batch: dst[m,:] = src[idx%T, idx/T, :] with
 * idx = flat_idx[m] -- the index map of rlpyt/algos/pg/ppo.py:94-95 applied by
 * namedarraytuple slicing (ppo.py:99-100).  elem_bytes = bytes per [t,b] row. */
int rlpyt_gather_tb(const void* src, const int64_t* flat_idx, void* dst, int T, int64_t B,
                    int64_t elem_bytes, int64_t M, rlpyt_stream_t stream);

/* Conv-stack input preparation in ONE pass: minibatch gather (flat_idx as above; NULL =
 * identity over M rows) + uint8 -> float32 * scale (rlpyt/models/pg/atari_ff_model.py:50-51)
 * + CHW -> HWC so the result is the channels-last storage of a logical [M,C,H,W] tensor.
 * src u8 [T*B, C, HW]; dst f32 [M, HW, C]. */
int rlpyt_obs_to_nhwc_f32(const uint8_t* src, const int64_t* flat_idx /*nullable*/, float* dst,
                          int T, int64_t B, int C, int64_t HW, int64_t M, float scale,
                          rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-time-step device work of the HBM-resident sampler.
 *
 * rlpyt_commit_rows: the row writes of rlpyt/samplers/parallel/gpu/collectors.py:30-47
 * (`env_buf.observation[t] = step.observation`, `agent_buf.action[t] = ...`, ...) for up to
 * 64 leaves in ONE launch.  Entry e copies nbytes from src to
 *     dst + (t + dt) * row_stride_bytes + col_off_bytes,
 * with t read from device memory (*t_dev, or 0 when t_dev is null) so a captured hipGraph
 * serves every time step.  `table_dev` is an array of rlpyt_row_copy in DEVICE memory;
 * max_entry_bytes sizes the grid.
 *
 * rlpyt_categorical_head_f32: policy/value heads + softmax + action sampling of
 * rlpyt/models/pg/atari_ff_model.py:56-58 and rlpyt/distributions/categorical.py:28-31
 * (sample_mode forward only): prob = softmax(h w_pi^T + b_pi) [n,A]; value = h w_v^T + b_v
 * [n] (w_v/value nullable together); action[i] = min{a : sum_{a'<=a} prob[i,a'] >
 * uniforms[i]} (inverse CDF; action nullable = no sampling).  With u_row_dev the uniforms
 * of this call are row *u_row_dev of a [T', n] table drawn once per batch, so a captured
 * hipGraph contains no RNG state.  A <= 32. */
typedef struct rlpyt_row_copy {
  void* dst;
  const void* src;
  int64_t row_stride_bytes;
  int64_t col_off_bytes;
  int64_t nbytes;
  int32_t dt;
  int32_t reserved;
  /* ABI 12: zero_where != NULL -> unit u = bytes [u * unit_bytes, (u + 1) * unit_bytes) of the entry is
   * written as zeros where zero_where[u] != 0 (wait-reset collector: blank rows of finished envs,
   * rlpyt/samplers/parallel/gpu/collectors.py:85-91) */
  const uint8_t* zero_where;
  int64_t unit_bytes;
} rlpyt_row_copy;
int rlpyt_commit_rows(const rlpyt_row_copy* table_dev, int n_entries, int64_t max_entry_bytes,
                      const int64_t* t_dev /*nullable*/, rlpyt_stream_t stream);
int rlpyt_categorical_head_f32(const float* h /*[n,K]*/, const float* w_pi /*[A,K]*/,
                               const float* b_pi, const float* w_v /*[K], nullable*/,
                               const float* b_v, const float* uniforms /*[n] or [T',n], nullable*/,
                               const int64_t* u_row_dev /*nullable: row *u_row_dev of uniforms*/,
                               int64_t n, int K, int A, float* prob, float* value,
                               int64_t* action, rlpyt_stream_t stream);

/* The sampler master's steady-state loop over time steps [t_begin, t_end) in native code --
 * the role of ActionServer.serve_actions (rlpyt/samplers/parallel/gpu/action_server.py:44-58)
 * once each pipeline group's per-step device work is a captured hipGraph.  Event-driven: each
 * group cycles on its own -- env workers arrived (obs_word reached rounds * n_workers) ->
 * enqueue the H2D copies of the page-locked step buffer (frame-stacked envs: newest frames +
 * the full stack of reset envs, t == 0: all full stacks), hipGraphLaunch, enqueue the D2H
 * action copies, record `event` -> event fired (hipEventQuery) -> rlpyt_seq_post(act_word) --
 * and the caller's thread services whichever hand-off is ready, so groups overlap freely and
 * may be at different time steps (each step's index reaches the device through t_host).
 * Returns RLPYT_ETIMEOUT after timeout_ms without any progress.  `acts` / `rounds`
 * are the running hand-off counters (updated in place).  timing[3] accumulates seconds spent
 * waiting for envs / issuing / waiting for the device.  Host pointers; blocks the caller. */
typedef struct rlpyt_copy_desc {
  void* dst;
  const void* src;
  int64_t nbytes;
} rlpyt_copy_desc;
typedef struct rlpyt_step_group {
  uint32_t* act_word;
  uint32_t* obs_word;
  uint32_t acts;
  uint32_t rounds;
  int32_t n_workers;
  int32_t n_h2d;
  rlpyt_copy_desc h2d[8];
  int32_t n_d2h;
  int32_t dedup;
  rlpyt_copy_desc d2h[4];
  int32_t Bg;
  int32_t reserved;
  const uint8_t* reset_flags; /* host [Bg], written by the workers */
  int32_t* slot_host;         /* host [Bg], part of the page-locked misc block */
  uint8_t* full_rows_dev;     /* device [Bg, row_bytes] */
  const uint8_t* obs_host;    /* host [Bg, row_bytes], page-locked */
  int64_t row_bytes;
  int64_t* t_host;  /* host, inside the page-locked misc block: receives the step index */
  void* graph_exec; /* hipGraphExec_t */
  void* stream;     /* hipStream_t */
  void* event;      /* hipEvent_t */
  /* tail pass (nullable): hipGraphExec_t run ONCE per group after step t_end - 1, behind the same
   * uploads with *t_host = t_end, when the workers' arrival after their last env step is in: the
   * bootstrap value / reward and done rows of the observation the batch ends on
   * (rlpyt/samplers/parallel/gpu/action_server.py:60-62).  No actions are published for it;
   * `rounds` advances by t_end - t_begin + 1 instead of t_end - t_begin. */
  void* tail_graph_exec;
} rlpyt_step_group;
int rlpyt_sampler_serve(rlpyt_step_group* groups, int n_groups, int t_begin, int t_end,
                        int spin_iters, int timeout_ms, double* timing /*[8], nullable*/);

/* Small-batch fully connected layer (+ optional ReLU) for the sampling forward -- the
 * 3456 -> 512 trunk of rlpyt/models/pg/atari_ff_model.py:52-55 (rlpyt/models/mlp.py) at
 * M <= 256 rows: y[m,n] = act(sum_k x[m,k] w[n,k] + bias[n]) on fp32 MFMA with the K range
 * split over workgroups and a fixed-order partial sum (deterministic).  x [M,K], w [N,K]
 * (torch nn.Linear layout), bias [N] nullable, y [M,N]; N % 16 == 0, K % 16 == 0;
 * workspace: rlpyt_fc_small_workspace_bytes(M, N) bytes; with y == NULL only the split-K
 * partials [ksplit, M, N] are left in workspace (consumed by rlpyt_lstm_cell_f32). */
int64_t rlpyt_fc_small_workspace_bytes(int M, int N);
int rlpyt_fc_small_ksplit(int K); /* number of K slices = leading dim of the partials */
int rlpyt_fc_small_f32(const float* x, const float* w, const float* bias /*nullable*/, float* y,
                       int M, int N, int K, int relu, float* workspace, rlpyt_stream_t stream);

/* Epsilon-greedy selection over Q-values for the DQN-family sampling step --
 * rlpyt/distributions/epsilon_greedy.py:17-29 (argmax, replaced by a uniformly random action with
 * probability epsilon) -- from one pre-drawn uniform per environment: action[b] =
 * u < eps ? floor(u / eps * A) : argmax_a q[b, a], u = uniforms[t_dev[0] * n + b] (t_dev NULL: row 0),
 * eps = eps[b * eps_stride] (stride 0: one epsilon for all, 1: vector epsilon).  q [n, A]. */
int rlpyt_eps_greedy_f32(const float* q, int64_t n, int A, const float* eps, int eps_stride,
                         const float* uniforms /*[T, n]*/, const int64_t* t_dev /*nullable*/,
                         int64_t* action /*[n]*/, rlpyt_stream_t stream);

/* Inputs of one recurrent sampling step in one launch (rlpyt/agents/dqn/r2d1_agent.py:23-40 + the
 * reset handling of rlpyt/samplers/parallel/gpu/action_server.py:49-53, rlpyt/agents/base.py:283-297):
 * xh [B,Kp] = [act(feat [B,F]) | onehot(action [B], A) | reward [B] | h [B,H] | 0-pad], with null action 0 /
 * zero reward / zero state for rows with done[b] (done nullable: no resets); prev_h, prev_c [B,H] = the
 * state the step starts from; c zeroed in place for reset rows.  relu != 0: act = ReLU. */
int rlpyt_rnn_step_inputs_f32(const float* feat, int F, int relu, const int64_t* action, int A,
                              const float* reward, const uint8_t* done /*nullable*/, const float* h,
                              float* c, int H, float* xh, int Kp, float* prev_h, float* prev_c,
                              int64_t B, rlpyt_stream_t stream);

/* Q-value head behind a split-K hidden layer (rlpyt/models/mlp.py:24-31 as the Linear-ReLU-Linear `head`
 * of rlpyt/models/dqn/atari_dqn_model.py:50-51 / atari_r2d1_model.py:44-45): partial = the split-K
 * partials [ksplit, n, K] of rlpyt_fc_small_f32(x, W_hidden) (y == NULL); q[n,A] = W_out relu(sum partial +
 * b_hidden) + b_out.  K in {256, 512}, A <= 18. */
int rlpyt_q_head_f32(const float* partial, int ksplit, const float* b_hidden, const float* w_out,
                     const float* b_out, int64_t n, int K, int A, float* q, rlpyt_stream_t stream);
/* The same head inside an update (round 6, ABI 16; autograd through `self.head` in the online network's pass of
 * DQN.loss, rlpyt/algos/dqn/dqn.py:226-230): rlpyt_q_head_train_f32 also keeps the hidden activations
 * h [n,K] (h_out nullable = rlpyt_q_head_f32); rlpyt_q_head_bwd_f32 turns dq [n,A] into dw_out [A,K] = dq^T h,
 * db_out [A], dh [n,K] = (dq w_out) * (h > 0) -- the gradient at the hidden layer's pre-activation -- and
 * db_hidden [K] = its column sums, in one launch (n <= 256, K % 64 == 0, A <= 18; fixed summation order).
 * The hidden layer's weight / input gradients (dh^T x, dh W_hidden) are GEMMs of the caller. */
int rlpyt_q_head_train_f32(const float* partial, int ksplit, const float* b_hidden, const float* w_out,
                           const float* b_out, int64_t n, int K, int A, float* q, float* h_out /*nullable*/,
                           rlpyt_stream_t stream);
int rlpyt_q_head_bwd_f32(const float* dq, const float* h, const float* w_out, int64_t n, int K, int A,
                         float* dw_out, float* db_out, float* dh, float* db_hidden, rlpyt_stream_t stream);

/* One LSTM cell step for the per-time-step sampling forward of the recurrent agents
 * (torch.nn.LSTM with T = 1 as used by rlpyt/models/dqn/atari_r2d1_model.py:61-63 and
 * rlpyt/models/pg/atari_lstm_model.py): the gate pre-activations arrive as the split-K partials
 * [ksplit, B, 4H] of rlpyt_fc_small_f32([x | h], [W_ih | W_hh]) (y == NULL), this adds b_ih + b_hh,
 * applies the gates in torch's order (i, f, g, o) and writes h' [B,H], c' [B,H]
 * (c' = sigmoid(f) c + sigmoid(i) tanh(g), h' = sigmoid(o) tanh(c')).  h_out / c_out may alias
 * c_prev's buffer only if they are the same element-for-element (in-place state update). */
int rlpyt_lstm_cell_f32(const float* partial, int ksplit, const float* b_ih, const float* b_hh,
                        const float* c_prev, float* h_out, float* c_out, int64_t B, int H,
                        rlpyt_stream_t stream);

/* Trunk + head of the rollout step -- the FC trunk of rlpyt/models/pg/atari_ff_model.py:52-55 and
 * rlpyt/agents/pg/categorical.py:34-43 + rlpyt/distributions/categorical.py:28-31:
 *   rlpyt_rollout_fc_f32: partial[s][m][n] = sum_{k in slice s} x[m,k] w[n,k], slices of 128 along
 *     K (ksplit = rlpyt_rollout_fc_ksplit(K) <= 32), one workgroup per (64 columns, slice, 64 rows);
 *     x [M,K], w [N,K], N % 64 == 0, K % 16 == 0, K <= 4096, M <= 1024; partial holds
 *     rlpyt_rollout_fc_workspace_bytes(M, N, K) bytes.  fp32 MFMA, fixed order: deterministic.
 *   rlpyt_rollout_head_f32: h = relu(sum_s partial[s] + fc_bias), policy / value heads + softmax +
 *     inverse-CDF draw (uniforms[t, row]) as rlpyt_categorical_head_f32, and the row writes of the
 *     step: prob_rows[t, lo+row, :], value_rows[t, lo+row], action_rows[t+1, lo+row],
 *     action_out[row] (may be a device-mapped address of the page-locked step buffer); t = *t_dev;
 *     one workgroup per row; with bootstrap_out != NULL it writes ONLY the value head's output to
 *     bootstrap_out[row] (the bootstrap value after the last step of a batch,
 *     rlpyt/samplers/parallel/gpu/action_server.py:60-62) and every row / uniform pointer may be
 *     NULL. */
int rlpyt_rollout_fc_ksplit(int K);
int64_t rlpyt_rollout_fc_workspace_bytes(int M, int N, int K);
int rlpyt_rollout_fc_f32(const float* x, const float* w, float* partial, int M, int N, int K,
                         rlpyt_stream_t stream);
int rlpyt_rollout_head_f32(const float* partial, int ksplit, const float* fc_bias,
                           const float* w_pi, const float* b_pi, const float* w_v,
                           const float* b_v, const float* uniforms /*[T', n]*/,
                           const int64_t* t_dev, int64_t n, int K, int A, float* prob_rows,
                           float* value_rows, int64_t* action_rows, int64_t B, int64_t lo,
                           int64_t* action_out, float* bootstrap_out /*nullable: [n]*/,
                           rlpyt_stream_t stream);

/* Frame-stack push for frame-stacked environments (rlpyt/envs/atari/atari_env.py:115-118:
 * the observation is the last C frames, newest last): the host uploads only the newest
 * frame of each env and row t of the HBM batch is rebuilt on the device,
 *   obs[t, lo+b] = slot[b] >= 0 ? full_rows[slot[b]]                (reset env / first step)
 *                               : concat(obs[t-1, lo+b, 1:], new_frame[b]),
 * t = *t_dev (t >= 1 wherever slot[b] < 0).  obs u8 [T,B,C,HW]; new_frame u8 [Bg,HW];
 * full_rows u8 [<=Bg,C,HW]; slot i32 [Bg]; stage u8 [Bg,C,HW] (nullable) receives a copy of
 * the rebuilt rows.  With reward_rows the step's scalar rows all_reward[t, lo:lo+Bg] and
 * all_done[t, lo:lo+Bg] are committed by the same launch.  Bit-exact byte moves. */
int rlpyt_frame_push(uint8_t* obs, const int64_t* t_dev, int64_t B, int64_t lo, int64_t Bg,
                     int C, int64_t HW, const uint8_t* new_frame, const uint8_t* full_rows,
                     const int32_t* slot, uint8_t* stage /*nullable*/,
                     float* reward_rows /*nullable: [T',B] f32*/, const float* reward_src /*[Bg]*/,
                     uint8_t* done_rows /*[T',B] bool*/, const uint8_t* done_src /*[Bg]*/,
                     rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * AtariFfModel convolution stack on fp32 MFMA -- rlpyt/models/pg/atari_ff_model.py:40-63 with
 * rlpyt/models/conv2d.py:8-117 at its default geometry: uint8 [4,104,80] -> conv(4->16, k8,
 * s4, p0) + ReLU -> conv(16->32, k4, s2, p1) + ReLU -> 3456 features (NCHW flatten order, so
 * the reference's nn.Linear(3456, 512) weight applies unchanged).  Fused in: the minibatch
 * gather idx -> (idx % T, idx / T) (flat_idx nullable = identity over M images, T/B then
 * only bound the row index), uint8 -> f32, the 1/255 scale, bias, ReLU and the backward
 * ReLU masks.  Weight layouts are torch's: w1 [16,4,8,8], w2 [32,16,4,4], contiguous.
 *   y1  f32 [M, 25*19, 16]  conv1 output after ReLU, channels-last (saved for backward)
 *   y2  f32 [M, 32*12*9]    conv2 output after ReLU
 *   g2  f32 [M, 3456]       dL/dy2;   dy1 f32 [M, 475, 16]  dL/d(conv1 pre-activation)
 * Weight gradients OVERWRITE dw / db (no accumulation); `workspace` holds
 * rlpyt_atari_conv_wgrad_workspace_bytes() bytes (per-workgroup partial sums, reduced in a
 * fixed order: deterministic). */
int rlpyt_atari_conv1_fwd_f32(const uint8_t* obs, const int64_t* flat_idx /*nullable*/, int T,
                              int64_t B, int64_t M, const float* w1, const float* b1,
                              float scale, float* y1, rlpyt_stream_t stream);
/* relu_mask u32 [M, 32, 4]: the sign bits of y2 (bit j of word [m][co][w] = y2[m][co][32 w + j] > 0),
 * written beside y2 -- all the backward pass needs of y2 (432 bits instead of 13.8 KB per image). */
int rlpyt_atari_conv2_fwd_f32(const float* y1, int64_t M, const float* w2, const float* b2,
                              float* y2, uint32_t* relu_mask, rlpyt_stream_t stream);
/* conv1 -> conv2 forward of an update minibatch in ONE pass over the images (round 6, ABI 9): the
 * two calls above with y1 handed from conv1 to conv2 through LDS -- y1 is still written (the backward
 * pass reads it) but never re-read: 636 MB instead of 888 MB per 8192 images.  Replaces the
 * `conv(img)` of rlpyt/models/pg/atari_ff_model.py:50-51 under autograd at update sizes; y1, y2 and
 * relu_mask are bit-identical to the two separate calls (same arithmetic, statement for statement).
 * M <= the number of CUs: forwards to the two latency-tuned launches. */
int rlpyt_atari_convs_fwd_f32(const uint8_t* obs, const int64_t* flat_idx /*nullable*/, int T,
                              int64_t B, int64_t M, const float* w1, const float* b1,
                              const float* w2, const float* b2, float scale, float* y1, float* y2,
                              uint32_t* relu_mask, rlpyt_stream_t stream);
/* Sampling-step front end in ONE launch (one environment per workgroup): the frame-stack push of
 * rlpyt_frame_push (obs[t, lo+b] rebuilt from slot / full_rows / obs[t-1] / new_frame, t = *t_dev,
 * optional reward/done row commit) followed by conv1 and conv2 of rlpyt_atari_conv{1,2}_fwd_f32 on
 * the rebuilt stack, which stays in LDS -- y1 never exists in HBM.  y2 f32 [Bg, 3456]; results
 * are bit-identical to the three separate launches (same accumulation order). */
int rlpyt_atari_sample_convs_f32(uint8_t* obs, const int64_t* t_dev, int64_t B, int64_t lo,
                                 int64_t Bg, const uint8_t* new_frame, const uint8_t* full_rows,
                                 const int32_t* slot, float* reward_rows /*nullable*/,
                                 const float* reward_src, uint8_t* done_rows,
                                 const uint8_t* done_src, const float* w1, const float* b1,
                                 const float* w2, const float* b2, float scale, float* y2,
                                 rlpyt_stream_t stream);
/* The same launch with the rebuilt stacks written to dst_stage [Bg, 4, 104, 80] instead of
 * obs[t, lo + b] (dst_stage NULL: identical to rlpyt_atari_sample_convs_f32): the bootstrap-value
 * pass on the observation AFTER the last step of a batch (t = T, obs[T-1] is still read, row T of
 * the batch does not exist). */
int rlpyt_atari_sample_convs_to_f32(uint8_t* obs, const int64_t* t_dev, int64_t B, int64_t lo,
                                    int64_t Bg, const uint8_t* new_frame,
                                    const uint8_t* full_rows, const int32_t* slot,
                                    float* reward_rows /*nullable*/, const float* reward_src,
                                    uint8_t* done_rows, const uint8_t* done_src, const float* w1,
                                    const float* b1, const float* w2, const float* b2, float scale,
                                    float* y2, uint8_t* dst_stage /*nullable*/,
                                    rlpyt_stream_t stream);
int64_t rlpyt_atari_conv_wgrad_workspace_bytes(void);
/* Importance-sampling weights of a prioritized batch (rlpyt/replays/non_sequence/prioritized.py:52-56,
 * sequence/prioritized.py:96-100) in one launch (round 6, ABI 13): out[i] = float(w_i / max_j w_j) with
 * w = (1 / (priorities + eps)) ** beta in float64; beta read from device memory when beta_dev != NULL
 * (captured update graphs), else the by-value argument.  n <= 65536. */
int rlpyt_is_weights_f64(const double* priorities, int64_t n, double eps, const double* beta_dev /*nullable*/,
                         double beta, float* out, rlpyt_stream_t stream);
/* No-grad forward of a single-layer LSTM over a sequence (torch.nn.LSTM in
 * rlpyt/models/dqn/atari_r2d1_model.py:61-63 as the target / warm-up / double-DQN passes of
 * rlpyt/algos/dqn/r2d1.py:199-224 run it).  xproj [T,B,4H] = x W_ih^T + b_ih + b_hh for every step
 * (torch gate order i,f,g,o; made by the caller with one GEMM); w_hh [4H,H]; h0 [B,H]; c [B,H]:
 * c0 on entry, c_T on return (updated in place); out [T,B,H] = h_1..h_T.  One launch per time
 * step; H in {256, 512}. */
int rlpyt_lstm_seq_f32(const float* xproj, const float* w_hh, const float* h0, float* c, float* out,
                       int T, int B, int H, rlpyt_stream_t stream);
/* The same sequence under autograd (round 6, ABI 11): the online network's training pass of
 * rlpyt/algos/dqn/r2d1.py:286-334 through the torch.nn.LSTM of atari_r2d1_model.py:61-63.
 * rlpyt_lstm_seq_train_f32 = rlpyt_lstm_seq_f32 that also keeps what the backward pass needs:
 *   gates [T,B,H,4] (activated i, f, g, o of every (step, sequence, unit)) and c_all [T,B,H] (c_1..c_T).
 * rlpyt_lstm_seq_bwd_f32 = back-propagation through time, one launch per step from T-1 to 0:
 *   dout [T,B,H] = dL/d out (nullable), dhn [B,H] = extra gradient into h_T (nullable),
 *   c0 [B,H] the initial cell state, w_hh_t [H,4H] = W_hh^T (made by the caller),
 *   dc [B,H]: dL/dc_T on entry (zeros if none), dL/dc_0 on return,
 *   dgates [T,B,4H] out: gradient w.r.t. the pre-activation gates in the module's order -- the caller
 *   takes dx = dgates W_ih, dW_ih = dgates^T x, dW_hh = dgates^T [h0; out_1..T-1], db = column sums;
 *   dh0 [B,H] out (nullable: one launch less).  Deterministic (fixed summation order). */
int rlpyt_lstm_seq_train_f32(const float* xproj, const float* w_hh, const float* h0, float* c, float* out,
                             float* gates, float* c_all, int T, int B, int H, rlpyt_stream_t stream);
int rlpyt_lstm_seq_bwd_f32(const float* dout, const float* dhn, const float* gates, const float* c_all,
                           const float* c0, const float* w_hh_t, float* dc, float* dgates, float* dh0,
                           int T, int B, int H, rlpyt_stream_t stream);

/* No-grad forward of the DQN-family conv stack with its default geometry -- Conv2d(4,32,8,s4) ReLU
 * Conv2d(32,64,4,s2,p1) ReLU Conv2d(64,64,3,s1,p1) ReLU, flattened: `self.conv` of
 * rlpyt/models/dqn/atari_dqn_model.py:30-37 and rlpyt/models/dqn/atari_r2d1_model.py:33-41 as
 * agent.step (agents/dqn/dqn_agent.py:61-68, r2d1_agent.py:40-53) and the target-network pass
 * (algos/dqn/dqn.py:226-234) run it.  obs: uint8 [N,4,104,80]; w1 [32,4,8,8], w2 [64,32,4,4],
 * w3 [64,64,3,3] in the torch layout, b* the biases; out: f32 [N, 64*12*9] in the order of
 * `conv(img).view(N, -1)`.  scale multiplies the pixels (1/255).  workspace: at least
 * rlpyt_dqn_convs_workspace_floats(N) floats (packed weights + the two intermediate layers).
 * packed == NULL: the weights are re-packed on the stream in front of the layers (four launches; always
 * current); packed != NULL: a copy made by rlpyt_dqn_convs_pack_f32 (rlpyt_dqn_convs_packed_floats()
 * floats) from the SAME weights -- three launches, w1 / w2 / w3 may then be NULL.  f32 MFMA, f32
 * accumulate. */
int64_t rlpyt_dqn_convs_workspace_floats(int64_t N);
int64_t rlpyt_dqn_convs_packed_floats(void);
int rlpyt_dqn_convs_pack_f32(const float* w1, const float* w2, const float* w3, float* packed,
                             rlpyt_stream_t stream);
int rlpyt_dqn_convs_fwd_f32(const uint8_t* obs, int64_t N, const float* w1, const float* b1,
                            const float* w2, const float* b2, const float* w3, const float* b3,
                            const float* packed /*nullable*/, float scale, float* workspace, float* out,
                            rlpyt_stream_t stream);

/* conv1 of that stack alone -- Conv2d(4, 32, 8, stride 4) + bias + ReLU on uint8 frames [N,4,104,80], y1
 * [N][475][32] channels-last -- as an exact bf16x3 contraction (the frames are exact in bf16, w1 split into
 * three bf16 pieces whose sum is exact; f32 accumulate): round 6, ABI 14.  rlpyt_dqn_convs_fwd_f32 runs its
 * first layer through it whenever w1 is given in the torch layout [32,4,8,8]. */
int rlpyt_dqn_conv1_f32(const uint8_t* obs, int64_t N, const float* w1, const float* b1, float scale,
                        float* y1, rlpyt_stream_t stream);

/* conv2 + conv3 of that stack on the bf16 matrix pipe ("bf16x6": both operands as three bf16 pieces, six
 * products of order <= 2, f32 accumulate; round 6, ABI 15): rlpyt_dqn_convs_x6_pack turns w2 [64,32,4,4] /
 * w3 [64,64,3,3] into operand-order pieces (rlpyt_dqn_convs_x6_packed_bytes() bytes), rlpyt_dqn_conv23_x6_f32
 * maps y1 [N][475][32] -> y2 [N][108][64] -> out [N][6912].  rlpyt_dqn_convs_pack_f32 / _fwd_f32 use them:
 * the packed buffer of rlpyt_dqn_convs_packed_floats() floats carries the pieces behind the f32 copies. */
int64_t rlpyt_dqn_convs_x6_packed_bytes(void);
int rlpyt_dqn_convs_x6_pack(const float* w2, const float* w3, void* packed, rlpyt_stream_t stream);
int rlpyt_dqn_conv23_x6_f32(const float* y1, int64_t N, const void* packed, const float* b2, const float* b3,
                            float* y2, float* out, rlpyt_stream_t stream);

/* Backward pass of the same stack at update-batch sizes (round 6, ABI 10): autograd through
 * `self.conv` of rlpyt/models/dqn/atari_dqn_model.py:30-37 in the online network's pass of DQN.loss
 * (rlpyt/algos/dqn/dqn.py:176-180).  y1 [N][475][32] and y2 [N][108][64] are the channels-last
 * activations rlpyt_dqn_convs_fwd_f32 left in its workspace (behind the packed weights), y3 = its
 * output [N][64*108], g3 = dL/dy3 in the same layout.  ReLU masks from the activations, bias
 * gradients = column sums; gradients OVERWRITE dw / db (torch layouts).  Weight packing + two data-
 * gradient kernels + three weight-gradient kernels + one fixed-order reduction of their per-image-
 * group partials (deterministic); f32 MFMA, f32 accumulate.  `workspace`:
 * rlpyt_dqn_convs_bwd_workspace_floats(N) floats, 16-byte aligned. */
int64_t rlpyt_dqn_convs_bwd_workspace_floats(int64_t N);
int rlpyt_dqn_convs_bwd_f32(const uint8_t* obs, int64_t N, const float* w2, const float* w3,
                            const float* y1, const float* y2, const float* y3, const float* g3,
                            float scale, float* workspace, float* dw1, float* db1, float* dw2,
                            float* db2, float* dw3, float* db3, rlpyt_stream_t stream);

/* conv2 backward in one pass (dgrad + both ReLU masks + weight / bias gradients; g2 / y1 read once,
 * conv2's ReLU mask from relu_mask as written by rlpyt_atari_conv2_fwd_f32), both contractions on
 * the bf16 matrix pipe (three-piece bf16 splits of both operands, six products, f32 accumulate:
 * f32-level error, like rlpyt_gemm_nt_f32). */
int rlpyt_atari_conv2_bwd_x6_f32(const float* g2, const uint32_t* relu_mask, const float* y1,
                                 int64_t M, const float* w2, float* dy1, float* workspace,
                                 float* dw2, float* db2, rlpyt_stream_t stream);
int rlpyt_atari_conv1_wgrad_f32(const uint8_t* obs, const int64_t* flat_idx /*nullable*/, int T,
                                int64_t B, int64_t M, const float* dy1, float scale,
                                float* workspace, float* dw1, float* db1, rlpyt_stream_t stream);

/* Generic 2-index gather: dst[m,:] = src[t_idx[m], b_idx[m], :] (negative t wraps once,
 * as numpy negative indexing does in rlpyt/replays/non_sequence/n_step.py:27-28). */
int rlpyt_gather_rows(const void* src, const int64_t* t_idx, const int64_t* b_idx, void* dst,
                      int T, int64_t B, int64_t elem_bytes, int64_t M, rlpyt_stream_t stream);

/* The small fields of a single-step replay batch in one launch -- NStepReturnBuffer.extract_batch,
 * rlpyt/replays/non_sequence/n_step.py:16-43 without the observations: for sample i at ring row
 * t = t_idx[i] (negative wraps once), column b = b_idx[i]:
 *   prev_action / prev_reward  = action / reward of row t - 1 (row -1 = T - 1), 0 where done[t - 1];
 *   out_action, out_return, out_done, out_done_n = row t of action / return_ / done / done_n;
 *   tgt_prev_action / tgt_prev_reward = action / reward of row (t + n_step) % T - 1, as stored.
 * Ring arrays [T, B]: action i64, reward / return_ f32, done / done_n u8 (bool).  Bit-exact moves. */
int rlpyt_replay_step_fields(const int64_t* action, const float* reward, const uint8_t* done,
                             const float* return_, const uint8_t* done_n, const int64_t* t_idx,
                             const int64_t* b_idx, int64_t n, int T, int64_t B, int n_step,
                             int64_t* prev_action, float* prev_reward, int64_t* out_action,
                             float* out_return, uint8_t* out_done, uint8_t* out_done_n,
                             int64_t* tgt_prev_action, float* tgt_prev_reward,
                             rlpyt_stream_t stream);

/* Replay append in one launch (ABI 17) -- BaseNStepReturnBuffer.append_samples' slice assignments
 * (rlpyt/replays/n_step.py:60-83: samples[idxs] = ...) and FrameBufferMixin.append_samples
 * (rlpyt/replays/frame.py:39-59) for T new time steps at ring row `start` of a [ring_T, B] ring,
 * T <= ring_T, rows wrapping at ring_T:
 *   field f:  ring_f[(start + t) % ring_T] = src_f[t]         (row_bytes = B * item bytes each);
 *   frames (nullable together with obs): frames u8 [ring_T + C - 1, B, frame_bytes], obs u8
 *     [T, B, C, frame_bytes]:  frames[C - 1 + (start + t) % ring_T, b] = obs[t, b, C - 1];
 *     start == 0: frames[f, b] = obs[0, b, f] for f < C - 1 (history of row 0);
 *     else if (start + T) % ring_T <= start (the lap closed): frames[j] = frames[ring_T + j] for
 *     j < C - 1, read AFTER the newest-frame writes (`<=`: DESIGN section 3's stated deviation
 *     from the strict `<` of frame.py:57 when an append of exactly ring_T rows lands on its start).
 * `fields` is a HOST array.  Pure byte moves, one writer per destination byte: bit-exact. */
typedef struct rlpyt_append_field {
  void* ring;
  const void* src;
  int64_t row_bytes;
} rlpyt_append_field;
int rlpyt_replay_append(const rlpyt_append_field* fields, int n_fields, const uint8_t* obs,
                        uint8_t* frames, int64_t frame_bytes, int C, int64_t T, int64_t B,
                        int64_t start, int64_t ring_T, rlpyt_stream_t stream);

/* NStepFrameBuffer.extract_observation -- rlpyt/replays/non_sequence/frame.py:14-30.
 * frames u8 [T+C-1, B, H*W]; done u8 [T,B]; obs[i,c,:] = frames[t_i+c, b_i, :], then for
 * f=1..C-1: if done[(t_i-f) mod T, b_i]: obs[i, :C-f] = 0. */
int rlpyt_frames_gather(const uint8_t* frames, const uint8_t* done, const int64_t* t_idx,
                        const int64_t* b_idx, uint8_t* obs /*[n,C,HW]*/, int64_t n, int T,
                        int64_t B, int C, int64_t HW, rlpyt_stream_t stream);

/* SequenceNStepFrameBuffer.extract_observation -- rlpyt/replays/sequence/frame.py:17-50.
 * obs [seq_T, n, C, HW]; wrap at T (head rows duplicated), post-reset blanking. */
int rlpyt_frames_gather_seq(const uint8_t* frames, const uint8_t* done, const int64_t* t_idx,
                            const int64_t* b_idx, uint8_t* obs, int64_t n, int seq_T, int T,
                            int64_t B, int C, int64_t HW, rlpyt_stream_t stream);

/* The two observation gathers of NStepReturnBuffer.extract_batch -- agent inputs at t_i and target
 * inputs at (t_i + n_step) mod T, rlpyt/replays/non_sequence/n_step.py:29-42 with
 * frame.py:14-30 for each -- in ONE launch: obs [2, n, C, HW], obs[0] = the stack at t_i,
 * obs[1] = the stack at t_i + n_step (same blanking rule for both). */
int rlpyt_frames_gather_pair(const uint8_t* frames, const uint8_t* done, const int64_t* t_idx,
                             const int64_t* b_idx, uint8_t* obs, int64_t n, int n_step, int T,
                             int64_t B, int C, int64_t HW, rlpyt_stream_t stream);

/* extract_sequences -- rlpyt/utils/misc.py:38-56: dst[s,i,:] = src[(t_i+s) mod T, b_i, :]. */
int rlpyt_gather_sequences(const void* src, const int64_t* t_idx, const int64_t* b_idx,
                           void* dst, int64_t n, int seq_T, int T, int64_t B,
                           int64_t elem_bytes, rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Sum tree (f64, HBM resident) -- rlpyt/replays/sum_tree.py:8-222.
 * Cursor / wrap-guard logic (sum_tree.py:60-99) runs on the host inside advance();
 * the tree itself never leaves HBM.
 * ---------------------------------------------------------------------------------- */
typedef struct rlpyt_sumtree rlpyt_sumtree;

int rlpyt_sumtree_create(rlpyt_sumtree** out, int T, int B, int off_backward, int off_forward,
                         double default_value, int enable_input_priorities,
                         int input_priority_shift);
void rlpyt_sumtree_destroy(rlpyt_sumtree* t);
int rlpyt_sumtree_reset(rlpyt_sumtree* t, rlpyt_stream_t stream);
/* geometry queries (host): */
int rlpyt_sumtree_levels(const rlpyt_sumtree* t);
int64_t rlpyt_sumtree_low_idx(const rlpyt_sumtree* t);
int rlpyt_sumtree_cursor(const rlpyt_sumtree* t);
/* device pointer to the f64 tree (2^levels - 1 nodes), for inspection / parity tests. */
double* rlpyt_sumtree_data(rlpyt_sumtree* t);
/* async device-to-device copy of the whole tree into dst (2^levels - 1 doubles). */
int rlpyt_sumtree_copy_tree(rlpyt_sumtree* t, double* dst, rlpyt_stream_t stream);

/* advance -- sum_tree.py:60-99,155-204.  `priorities` (device f64, nullable): kind
 * 0 = none (default value), 1 = scalar [1], 2 = [B], 3 = [T_new,B]. */
int rlpyt_sumtree_advance(rlpyt_sumtree* t, int T_new, const double* priorities, int kind,
                          rlpyt_stream_t stream);
/* sample/find -- sum_tree.py:101-128,211-222.  uniforms: device f64 [n] in [0,1).
 * Outputs (device): T_idxs, B_idxs i64 [n]; priorities f64 [n] (nullable);
 * The sampled tree indices are remembered inside the handle (device) for update(). */
int rlpyt_sumtree_sample(rlpyt_sumtree* t, const double* uniforms, int n, int64_t* T_idxs,
                         int64_t* B_idxs, double* priorities, rlpyt_stream_t stream);
/* Last step of sample(n, unique=True) -- sum_tree.py:109-128: `leaves` (device int64 [n], leaf =
 * T_idx * B + B_idx) become the set the next rlpyt_sumtree_update applies to; their priorities are
 * returned (nullable).  The de-duplication / re-draw loop itself runs in the binding, on the host
 * RNG stream the reference uses. */
int rlpyt_sumtree_set_sampled(rlpyt_sumtree* t, const int64_t* leaves, int n, double* priorities,
                              rlpyt_stream_t stream);
/* update_batch_priorities -- sum_tree.py:130-153,206-209.  new_priorities device f64 [n]
 * (already ** alpha); duplicates among the last sampled indices are removed keeping the
 * FIRST occurrence in batch order (np.unique(return_index=True) semantics). */
int rlpyt_sumtree_update(rlpyt_sumtree* t, const double* new_priorities, int n,
                         rlpyt_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Gradient clipping + Adam for a whole model in two launches (multi-tensor).
 * Replaces, per minibatch update, the host sequence of rlpyt/algos/pg/ppo.py:100-104
 * (a2c.py:52-56, dqn/dqn.py:176-180, dqn/r2d1.py):
 *     grad_norm = torch.nn.utils.clip_grad_norm_(agent.parameters(), clip_grad_norm)
 *     optimizer.step()            -- torch.optim.Adam (rlpyt/algos/pg/base.py:34-37)
 * `tensors_host`: HOST array of n_tensors (<= RLPYT_ADAM_MAX_TENSORS) records of DEVICE
 * pointers (param, grad, exp_avg, exp_avg_sq; f32; 16-byte aligned tensors take the vector
 * path, others a scalar one) and element counts; read
 * during the call only.  `step` is the 1-based update count (bias corrections are formed on the
 * host in double).  max_norm <= 0 disables clipping.  `grad_norm_out` (nullable, device, 1 float)
 * receives the total 2-norm BEFORE clipping, as clip_grad_norm_ returns it.  The gradient
 * tensors are not modified (the clip coefficient is applied on the fly).
 * `workspace`: >= rlpyt_clip_adam_workspace_bytes() bytes of device scratch. */
#define RLPYT_ADAM_MAX_TENSORS 32
typedef struct rlpyt_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} rlpyt_adam_tensor;
int64_t rlpyt_clip_adam_workspace_bytes(void);
int rlpyt_clip_adam_step_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors, double lr,
                             double beta1, double beta2, double eps, double weight_decay,
                             int64_t step, double max_norm, void* workspace,
                             float* grad_norm_out, rlpyt_stream_t stream);
/* Variants for CAPTURED minibatch updates (one hipGraph replayed for every update of the epochs x
 * minibatches loop of rlpyt/algos/pg/ppo.py:92-104):
 *  rlpyt_update_tick -- first launch of an update: cur = *ctr (clamped to the table);
 *    hyper_cur[0:n_cols] = table[cur]; idx_static[0:M] = idx_all[cur*M : (cur+1)*M] (the update's
 *    minibatch indices, at the fixed address the captured kernels read; both nullable together);
 *    tick_idx[0] = cur (nullable).  table [n_rows, n_cols] f32 is computed on the host per
 *    iteration (row k: lr / bc1, 1 / sqrt(bc2) of the k-th update, then free columns, e.g. the
 *    ratio clip) -- the same double-precision bias corrections as the eager path, bit for bit.
 *  rlpyt_clip_adam_step_dev_f32 -- rlpyt_clip_adam_step_f32 with hyper_dev (nullable, device:
 *    {lr / bc1, 1 / sqrt(bc2)}) overriding lr / step at run time, and tick_ctr (nullable) advanced
 *    by one at the end of the update. */
int rlpyt_update_tick(const int64_t* ctr, const float* table, int n_rows, int n_cols,
                      float* hyper_cur, const int64_t* idx_all /*nullable*/,
                      int64_t* idx_static /*nullable*/, int64_t M, int64_t* tick_idx /*nullable*/,
                      rlpyt_stream_t stream);
int rlpyt_clip_adam_step_dev_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors, double lr,
                                 double beta1, double beta2, double eps, double weight_decay,
                                 int64_t step, double max_norm, void* workspace,
                                 float* grad_norm_out, const float* hyper_dev /*nullable*/,
                                 int64_t* tick_ctr /*nullable*/, rlpyt_stream_t stream);
/* rlpyt_clip_adam_step_dev_f32 that ALSO keeps a transposed copy of one parameter current: tensor
 * `mirror_index` of the table is a row-major [mirror_rows, mirror_cols] matrix (both multiples of 32,
 * 16-byte aligned) and `mirror_pt` [mirror_cols, mirror_rows] receives its new values transposed, from
 * the launch that writes them (32 x 32 tiles through LDS).  For the update trunk's weight W of
 * rlpyt/models/mlp.py:24-31: its input-gradient GEMM g W reads W^T (rlpyt_gemm_nt_f32 on the
 * transposed operand), which was a 7 MB transposing copy per minibatch.  mirror_index < 0: no mirror
 * (= rlpyt_clip_adam_step_dev_f32). */
int rlpyt_clip_adam_step_mirror_f32(const rlpyt_adam_tensor* tensors_host, int n_tensors, double lr,
                                    double beta1, double beta2, double eps, double weight_decay,
                                    int64_t step, double max_norm, void* workspace,
                                    float* grad_norm_out, const float* hyper_dev /*nullable*/,
                                    int64_t* tick_ctr /*nullable*/, int mirror_index,
                                    float* mirror_pt /*nullable*/, int64_t mirror_rows,
                                    int64_t mirror_cols, rlpyt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RLPYT_HIP_H */
