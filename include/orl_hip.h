/*
 * orl_hip.h - C ABI of the MI355X-native on-policy rollout + PPO update engine.
 *
 * The reference (OpenRL-Lab/openrl v0.2.1) has no native layer and therefore no FFI to be
 * compatible with; its seams are Python classes.  Every entry point below
 * replaces one group of torch/numpy ops on the reference hot path and cites it.  The host-side
 * mirror of the reference's Python interface (openrl_amd/) binds these symbols with ctypes; a
 * maintainer of the reference would bind them the same way.
 *
 * Conventions (all entry points):
 *   - extern "C"; plain pointers and sizes only, no torch / STL types;
 *   - return 0 on success, a negative ORL_E_* for an invalid argument, a positive value is a
 *     hipError_t from the launch; orl_last_error_string() describes the last failure (thread local);
 *   - every data pointer is a DEVICE pointer to contiguous float32 unless stated; the caller owns
 *     all memory; the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream, or NULL);
 *     launches are asynchronous and the functions are re-entrant.
 *
 * Buffer geometry (openrl/buffers/replay_data.py:41-184):
 *   T = episode_length, N = n_rollout_threads (envs), A = num_agents, L = N*A "lanes",
 *   arrays are [T(+1), N, A, width] C-contiguous; flat sample row of (t,n,a) = (t*N+n)*A+a.
 */
#ifndef ORL_HIP_H
#define ORL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORL_VERSION 306 /* 0.3.6: orl_tower_split_terms (the MLP towers' 64-wide GEMMs as two-term fp16 splits); 0.3.5: orl_build_experiments; orl_rollout_args.opp_reserved selects the rollout kernel of the single-agent device envs; the GEMM bits of orl_ppo_hparams.reserved are as round 5 re-encoded them for orl_rnn_ppo_fwd_bwd (0 = the production row kernels, 4 = recompute kernel, 8 | 16 = streamed kernels) - undefined combinations are rejected, and the comparison builds behind 4 (orl_ppo_fwd_bwd) / 8 exist only in an ORL_BUILD_EXPERIMENTS library; 0.3.4: orl_gt_train; 0.3.3: orl_gt_* (cross-layer fused general towers); 0.3.2: orl_gen_rollout_fused, ORL_HEAD_MIXED, hparams.reserved & 4; 0.3.0: orl_comm_error_copy; 0.2.6: orl_gen_act; 0.2.5: orl_gen_lstm_gate_fwd / _bwd, orl_gen_matmul, orl_gen_colsum_rows; 0.2.4: orl_gen_gru_gate_fwd / _bwd, orl_gen_row_affine, orl_gen_mlp_fwd feats; 0.2.3: orl_gen_layer_fwd / _bwd / orl_gen_wgrad / orl_gen_colsum; 0.2.2: orl_rollout_args.opp_per_reset..; 0.2.1: orl_rnn_rollout_fused; 0.2.0: rng_step_dev, struct sizes, collectives */

#define ORL_E_INVALID (-1)     /* bad size / null pointer / unsupported combination */
#define ORL_E_UNSUPPORTED (-2) /* legal in the reference, not built here (says which) */

/* ---- network description -------------------------------------------------------------------
 * One MLP "tower": [feature LayerNorm(D)] -> Linear(D,H) -> ReLU -> LayerNorm(H)
 *                  -> Linear(H,H) -> LayerNorm(H) -> head Linear(H,n_out)
 * = MLPBase/MLPLayer with layer_N = 1 (openrl/modules/networks/utils/mlp.py:8-46,100-180) followed
 * by ACTLayer's Categorical / DiagGaussian linear (utils/act.py:14-25, utils/distributions.py:58-98)
 * or ValueNetwork.v_out (networks/value_network.py:103-109).
 * `theta` is ONE flat float32 buffer in the reference's state_dict order:
 *   W1[H*D] b1[H] g1[H] be1[H] W2[H*H] b2[H] g2[H] be2[H] W3[n_out*H] b3[n_out] (logstd[n_out] if Gaussian)
 * All weight matrices are row-major [out][in] exactly as torch.nn.Linear stores them.
 */
#define ORL_HEAD_VALUE 0       /* critic: n_out = 1 */
#define ORL_HEAD_CATEGORICAL 1 /* Discrete(n_out) */
#define ORL_HEAD_GAUSSIAN 2    /* Box(n_out): mean head + state-independent logstd */

typedef struct orl_net_desc {
  int32_t obs_dim;   /* D */
  int32_t hidden;    /* H, must be 64 */
  int32_t n_out;     /* head width */
  int32_t head_kind; /* ORL_HEAD_* */
} orl_net_desc;

/* ABI guard for bindings: sizeof() of the structs of this header AS THE LIBRARY WAS BUILT, so that a stale shared
 * object with the same symbol names but another struct layout is refused at load time instead of corrupting device
 * memory.  which: 0 orl_net_desc, 1 orl_pack_src, 2 orl_buffer_ptrs, 3 orl_copy_desc, 4 orl_gather_desc,
 * 5 orl_ppo_hparams, 6 orl_adam_state, 7 orl_rollout_args, 8 orl_rnn_batch, 9 orl_rnn_rollout_args, 10 orl_gen_mlp_desc; anything else
 * returns ORL_E_INVALID. */
int orl_abi_struct_size(int which);

/* number of float32 parameters of a tower (== sum of reference state_dict numels) */
int orl_param_count(const orl_net_desc* net);

int orl_version(void);
const char* orl_last_error_string(void);

/* ---- K6 + K7a: GAE / return reverse scan -----------------------------------------------------
 * Replaces ReplayData.compute_returns (openrl/buffers/replay_data.py:320-423), all four variants.
 * flags: bit0 use_gae, bit1 use_proper_time_limits.
 * vn_state: NULL, or the 3 ValueNorm state floats {running_mean, running_mean_sq, debiasing_term}
 *   (openrl/modules/utils/valuenorm.py:24-35); when given, value_preds are de-normalised with
 *   ValueNorm.denormalize (valuenorm.py:93-106) exactly where the reference does.
 * next_value [L] is stored into value_preds[T] (use_gae) or returns[T] (otherwise) first.
 * Optional fused outputs (K7a, openrl/algorithms/ppo.py:384-400): adv_raw [T,L] =
 *   returns[:-1] - denorm(value_preds[:-1]) and stat_partials [gridDim][8] doubles holding per-block
 *   {sum, sumsq, count} over ALL entries and over entries with active_masks != 0, plus
 *   {sum, sumsq} of returns[:-1] (ValueNorm batch moments); pass NULL to skip.
 * *n_partials receives the number of partial rows written (host int, may be NULL).
 */
int orl_gae_scan(const float* rewards, float* value_preds, const float* masks, const float* bad_masks,
                 const float* next_value, const float* vn_state, float* returns, int T, int L,
                 double gamma, double gae_lambda, int flags, const float* active_masks, float* adv_raw,
                 double* stat_partials, int* n_partials, void* stream);
int orl_gae_max_partials(int T, int L);

/* K7a stand-alone (openrl/algorithms/ppo.py:384-400): adv_raw[T*L] = returns[:-1] - denorm(value_preds[:-1])
 * and the same per-block statistics rows as orl_gae_scan, for callers that did not take its fused outputs
 * (e.g. returns were edited after compute_returns).  stat_partials holds orl_gae_max_partials(T, L) rows. */
int orl_adv_stats(const float* returns, const float* value_preds, const float* active_masks,
                  const float* vn_state, int T, int L, float* adv_raw, double* stat_partials, int* n_partials,
                  void* stream);

/* ---- K7b: advantage normalisation (+ record packing) -------------------------------------------
 * Replaces openrl/algorithms/ppo.py:402-409: optional global (x-mean)/(std+1e-5) (use_adv_normalize)
 * followed - always - by the same transform with nanmean/nanstd over entries whose active mask != 0.
 * Reads the partials of orl_gae_scan; writes adv [T*L] in place over adv_raw (may alias).
 * stats_out (device, 11 doubles, optional): the 8 reduced sums, then {sum ret, sum ret^2, count} = the moments
 * orl_valuenorm_update takes when one minibatch is the whole batch (no extra pass, no gather).
 * When `records` != NULL also packs the per-sample update record (see orl_record_width):
 *   [policy_obs Dp | critic_obs Dc | action a | old_logp a | adv | value_pred | return | active | action_mask K]
 * in flat row order (t*N+n)*A+a, i.e. the row order of feed_forward_generator
 * (openrl/buffers/replay_data.py:594-613).
 */
typedef struct orl_pack_src {
  const float* policy_obs;       /* [T+1, L, Dp] */
  const float* critic_obs;       /* [T+1, L, Dc] */
  const float* actions;          /* [T, L, a]    */
  const float* action_log_probs; /* [T, L, a]    */
  const float* value_preds;      /* [T+1, L, 1]  */
  const float* returns;          /* [T+1, L, 1]  */
  const float* active_masks;     /* [T+1, L, 1]  */
  const float* action_masks;     /* [T+1, L, K] or NULL */
  int32_t Dp, Dc, a, K;
} orl_pack_src;

int orl_record_width(int Dp, int Dc, int a, int K); /* floats per record, multiple of 4 */
int orl_adv_normalize_pack(float* adv, const double* stat_partials, int n_partials, int T, int L,
                           int use_adv_normalize, double* stats_out, const orl_pack_src* src,
                           float* records, void* stream);

/* ---- K5: buffer insert + mask construction ----------------------------------------------------
 * Replaces OnPolicyDriver.add2buffer mask logic (openrl/drivers/onpolicy_driver.py:91-138) and
 * ReplayData.insert (openrl/buffers/replay_data.py:245-284) for one rollout step `step`:
 *   policy_obs/critic_obs[step+1] <- next obs, rewards[step] <- rewards,
 *   masks[step+1] = 0 where ALL agents of the env are done else 1,
 *   active_masks[step+1] = 0 for a done agent unless the whole env is done, else 1,
 *   bad_masks[step+1] = 0 where bad_transition else 1, action_masks[step+1] <- next masks (optional).
 * dones / bad_transition are uint8 [N, A]; bad_transition and next_action_masks may be NULL.
 */
typedef struct orl_buffer_ptrs {
  float* policy_obs;   /* [T+1, N, A, Dp] */
  float* critic_obs;   /* [T+1, N, A, Dc] (may alias policy_obs) */
  float* rewards;      /* [T, N, A, 1] */
  float* masks;        /* [T+1, N, A, 1] */
  float* bad_masks;    /* [T+1, N, A, 1] */
  float* active_masks; /* [T+1, N, A, 1] */
  float* action_masks; /* [T+1, N, A, K] or NULL */
  int32_t T, N, A, Dp, Dc, K;
} orl_buffer_ptrs;

int orl_buffer_insert(const orl_buffer_ptrs* buf, int step, const float* next_policy_obs,
                      const float* next_critic_obs, const float* rewards, const uint8_t* dones,
                      const uint8_t* bad_transition, const float* next_action_masks, void* stream);
/* orl_buffer_insert for recurrent policies: additionally multiplies the hidden states the act step left in slot
 * step+1 (h_*_next = rnn_states[step+1], [N*A, hidden], in place; h_critic_next may be NULL) by masks[step+1], i.e.
 * rnn_states[dones_env == True] = 0 (onpolicy_driver.py:100-109) without two extra elementwise launches. */
int orl_buffer_insert_rnn(const orl_buffer_ptrs* buf, int step, const float* next_policy_obs,
                          const float* next_critic_obs, const float* rewards, const uint8_t* dones,
                          const uint8_t* bad_transition, const float* next_action_masks, float* h_policy_next,
                          float* h_critic_next, int hidden, void* stream);

/* ReplayData.after_update (openrl/buffers/replay_data.py:286-318): dst[k][0..n[k]) = src[k][0..n[k]) for up to 8
 * float arrays in one launch (slot T -> slot 0 of the per-step arrays). */
#define ORL_COPY_MAX 8
typedef struct orl_copy_desc {
  const float* src[ORL_COPY_MAX];
  float* dst[ORL_COPY_MAX];
  int64_t n[ORL_COPY_MAX];
  int32_t count;
} orl_copy_desc;
int orl_multi_copy(const orl_copy_desc* desc, void* stream);

/* ---- K8: minibatch gather ------------------------------------------------------------------------
 * Replaces the fancy-index gathers of ReplayData.feed_forward_generator
 * (openrl/buffers/replay_data.py:615-646): dst[i, :] = src[idx[i], :] for up to 12 arrays at once.
 * idx is int64 (the dtype torch.randperm produces); rows are `width` floats.
 */
#define ORL_GATHER_MAX 12
typedef struct orl_gather_desc {
  const float* src[ORL_GATHER_MAX];
  float* dst[ORL_GATHER_MAX];
  int32_t width[ORL_GATHER_MAX];
  int32_t count;
} orl_gather_desc;
int orl_gather_minibatch(const orl_gather_desc* desc, const int64_t* idx, int n_rows, void* stream);

/* Keyed pseudo-random permutation of [0, n): idx[i] = cycle-walking 4-round Feistel over
 * ceil(log2 n) bits keyed by (seed, stream).  Engine-native alternative to the host
 * torch.randperm used by BatchSampler(SubsetRandomSampler) (replay_data.py:578-580); NOT
 * bit-identical to it (documented in DESIGN.md). */
int orl_perm_feistel(int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* Same permutation plus, in the same launch, orl_valuenorm_update(vn_state, moments, beta): the per-epoch pair when
 * the minibatch is the whole batch (num_mini_batch = 1: moments come from the GAE pass, ppo.py:190-195). */
int orl_perm_feistel_vn(int64_t* idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn_state,
                        const double* moments, double beta, void* stream);

/* ---- K1-K4: rollout forward + sampling -------------------------------------------------------------
 * Replaces PPOModule.get_actions (openrl/modules/ppo_module.py:102-138) =
 * PolicyNetwork.forward_original (networks/policy_network.py:130-162) + ValueNetwork.forward
 * (networks/value_network.py:113-136) + ACTLayer.forward (utils/act.py:45-83) for B rows.
 *   values  [B,1]   = critic(critic_obs)         (NULL critic theta -> skipped; NULL policy theta ->
 *                                                  value-only call = PPOModule.get_values, :140-147)
 *   actions [B,a]   sampled (or mode if deterministic) as float32, logp [B,a]
 * Categorical: logits[mask==0] = -6e4 (distributions.py:71), a = 1, action = index as float.
 * Gaussian: per-dimension log-prob, no sum (distributions.py:34-43), a = n_out.
 * Sampling uses Philox4x32-10 keyed by `seed`, counter (row0 + row, rng_step): u = top 24 bits;
 * categorical = inverse CDF over softmax probabilities; normal = Box-Muller.  If `forced_u` != NULL
 * its [B, a] uniforms (categorical) / standard normals (Gaussian) replace the generator
 * (teacher-forced parity tests).
 */
int orl_act_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet,
                 const float* ctheta, const float* policy_obs, const float* critic_obs,
                 const float* action_masks, int B, int deterministic, uint64_t seed, uint64_t row0,
                 uint64_t rng_step, const uint64_t* rng_step_dev, const float* forced_u, float* values,
                 float* actions, float* logp, void* stream);
/* orl_act_step (policy only) for a POOL of policies of one architecture in one launch: rows
 * [g*rows_per_group, (g+1)*rows_per_group) are evaluated with the parameters pthetas + g*theta_stride
 * (rows_per_group a multiple of 16).  The self-play env uses it for the opponents' moves. */
int orl_act_step_grouped(const orl_net_desc* pnet, const float* pthetas, int64_t theta_stride, int rows_per_group,
                         const float* policy_obs, const float* action_masks, int B, int deterministic, uint64_t seed,
                         uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev, float* actions, float* logp,
                         void* stream);

/* orl_act_step (policy only) for a pool of K policies with a PER-ROW assignment: row i is evaluated with the
 * parameters pthetas + opp_index[i]*theta_stride.  A 16-row tile runs the tower once per distinct index it contains
 * (exact for any assignment; K x the work in the worst case) - the opponents of the self-play env when every env
 * draws its opponent at reset (openrl/selfplay/wrappers/opponent_pool_wrapper.py:37-66). */
int orl_act_step_pool(const orl_net_desc* pnet, const float* pthetas, int64_t theta_stride, int n_policies,
                      const int32_t* opp_index, const float* policy_obs, const float* action_masks, int B,
                      int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                      float* actions, float* logp, void* stream);
/* Opponent sampling of the self-play pool (openrl/selfplay/sample_strategy/{random,last}_opponent.py): for every env n
 * with dones[n] != 0 (dones == NULL: every env) opp_index[n] <- strategy 0: uniform over [0, n_filled) from Philox
 * keyed (seed, n, draw_id); strategy 1: last_slot (the newest snapshot).  per_tile != 0: one draw per 16-env tile,
 * written to all of its envs (what the fused rollout kernel can consume). */
int orl_opponent_sample(int32_t* opp_index, const uint8_t* dones, int N, int n_filled, int last_slot, int strategy,
                        int per_tile, uint64_t seed, uint64_t draw_id, const uint64_t* draw_id_dev, void* stream);

/* `rng_step_dev` (orl_act_step / orl_act_step_grouped / orl_rnn_act_step; may be NULL): optional DEVICE-side addend
 * of `rng_step`: rng_step_effective = rng_step + *rng_step_dev, read by the kernel at run time.  It exists so that a
 * stepwise rollout can be captured ONCE into a hipGraph (kernel arguments are frozen at capture) and replayed every
 * iteration with fresh Philox counters - the caller advances the counter on the device.  An explicit argument: the
 * library keeps no per-thread or global state besides the last-error string. */

/* Forward-only evaluation of GIVEN actions = PPOModule.evaluate_actions (openrl/modules/ppo_module.py:149-193,
 * PolicyNetwork.eval_actions networks/policy_network.py:164-203, ACTLayer.evaluate_actions utils/act.py:102-172):
 * values [B,1], action_log_probs [B,a], dist_entropy [1] (device) = masked mean with active_masks [B,1] when
 * given, else the plain mean; entropy_rows [B] is scratch (entropy * weight per row).  ctheta may be NULL. */
int orl_evaluate_actions(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet,
                         const float* ctheta, const float* policy_obs, const float* critic_obs,
                         const float* actions, const float* action_masks, const float* active_masks, int B,
                         float* values, float* action_log_probs, float* entropy_rows, float* dist_entropy,
                         void* stream);

/* ---- K9-K12: fused PPO minibatch forward + loss + backward ----------------------------------------
 * Replaces PPOAlgorithm.prepare_loss + loss.backward() x2 (openrl/algorithms/ppo.py:238-361,
 * 123-124) with PPOModule.evaluate_actions (modules/ppo_module.py:149-193), cal_value_loss
 * (ppo.py:178-220) and huber/mse (modules/utils/util.py:20-27) for the rows idx[0..mb) of
 * `records`.  Produces UNNORMALISED gradient sums (per-workgroup partials, reduced by
 * orl_ppo_finalize) so that multi-GPU runs can all-reduce sums and denominators exactly.
 */
typedef struct orl_ppo_hparams {
  float clip_param;
  float entropy_coef;
  float value_loss_coef;
  float huber_delta;
  float dual_clip_coeff;
  float max_grad_norm;
  int32_t use_clipped_value_loss;
  int32_t use_huber_loss;
  int32_t use_value_active_masks;
  int32_t use_policy_active_masks;
  int32_t use_valuenorm; /* normalise returns with vn_state inside the value loss */
  int32_t dual_clip_ppo;
  int32_t use_max_grad_norm;
  int32_t reserved; /* flag bits: 1 = critic-only update (turn_on == False, ppo.py:226-236);
                     * 2 = A2C policy loss -adv*logp instead of the clipped surrogate (algorithms/a2c.py:88-98);
                     * 4 = orl_ppo_fwd_bwd forms every GEMM on v_mfma_f32_16x16x4_f32 instead of the bf16x3 split (a
                     *     measurement / comparison switch: same results within fp32 rounding, ~1.4x slower);
                     *     orl_rnn_ppo_fwd_bwd: the row kernel with forward recompute (rnn_row_pair_kernel) whatever the chunk
                     *     length - comparison switch; 0 = the default: chunks of 2 steps run rnn_row2_pair_kernel (both steps
                     *     resident in registers), any other length rnn_row_pair_kernel;
                     * 8 = orl_ppo_fwd_bwd skips the transposing-read full-split build (dgrad through
                     *     ds_read_b64_tr_b16 of W2's bf16 image, the wide-observation towers' default) and takes round 3's
                     *     variants (two images, or wgrad-only split) - comparison switch, same arithmetic;
                     *     orl_rnn_ppo_fwd_bwd: round 4's streamed bf16-split row kernel - comparison switch;
                     * 16 = orl_rnn_ppo_fwd_bwd with 8: the streamed kernel with 4 waves per workgroup (one wave per SIMD,
                     *     512 registers) instead of 8 - comparison switch, same arithmetic;
                     * 32 = orl_ppo_apply* OVERWRITES its train_info slots instead of adding to them (the first optimiser step
                     *     of a train() call: saves the caller a zero-fill launch) */
} orl_ppo_hparams;

/* size (floats) of the raw gradient-sum vector of one tower and of the stats vector */
int orl_raw_grad_count(const orl_net_desc* net);
#define ORL_N_STATS 16
int orl_ppo_max_blocks(void); /* upper bound of workgroups => rows of `partials` */

/* partials: float32 device scratch of orl_ppo_max_blocks() * (raw_p + ORL_N_STATS + raw_c + ORL_N_STATS)
 * floats: policy rows [n_blocks][raw_p + ORL_N_STATS] start at 0, critic rows
 * [n_blocks][raw_c + ORL_N_STATS] start at orl_ppo_max_blocks() * (raw_p + ORL_N_STATS).
 * n_blocks_out (host int[2], optional) receives the rows written to the two regions.
 * idx may be NULL (identity order).  vn_state is read AFTER the caller applied
 * orl_valuenorm_update for this minibatch (ppo.py:190-195 order). */
int orl_ppo_fwd_bwd(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet,
                    const float* ctheta, const float* records, int rec_width, const int64_t* idx,
                    int mb, const float* vn_state, const orl_ppo_hparams* hp, float* partials,
                    int* n_blocks_out, void* stream);

/* ---- K13 + K14 (+ ValueNorm state, stats) -----------------------------------------------------------
 * orl_ppo_reduce: sums the per-workgroup partials into `sums` [raw_p + raw_c + ORL_N_STATS]
 *   (this is the vector a multi-GPU run all-reduces - SURVEY.md section 8e).
 * orl_ppo_apply: turns the summed raw vector into parameter gradients (divides by the masked-mean
 *   denominators), computes both global grad norms, clips (torch clip_grad_norm_,
 *   ppo.py:132-145), runs one torch.optim.Adam step per tower (modules/rl_module.py:80-85;
 *   lr/eps/weight_decay as given, betas (0.9, 0.999)) and writes the six train_info scalars
 *   {value_loss, policy_loss, dist_entropy, actor_grad_norm, critic_grad_norm, ratio} (ppo.py:445-451)
 *   to train_info_accum[0..6) by ADDING (so an epoch loop accumulates like ppo.py:445-451).
 */
typedef struct orl_adam_state {
  float* theta; /* parameters, updated in place */
  float* grad;  /* gradient out (parameter order), for inspection */
  float* m;     /* exp_avg */
  float* v;     /* exp_avg_sq */
  float lr, eps, weight_decay;
  int32_t step; /* 1-based step count of THIS update */
} orl_adam_state;

int orl_ppo_reduce(const float* partials, int n_blocks, int width, float* sums, void* stream);
/* Same, both regions of an orl_ppo_fwd_bwd `partials` buffer in ONE launch:
 * sums[0 .. width_policy) <- policy region, sums[width_policy .. +width_critic) <- critic region. */
int orl_ppo_reduce_pair(const float* partials, int n_blocks_policy, int width_policy, int n_blocks_critic,
                        int width_critic, float* sums, void* stream);
int orl_ppo_apply(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums,
                  const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                  float* train_info_accum, void* stream);
/* orl_ppo_apply plus, on otherwise idle workgroups of the same launch, the NEXT epoch's minibatch permutation
 * (orl_perm_feistel(next_idx, n, seed, stream_id)) and - when vn_state != NULL - its ValueNorm.update
 * (orl_valuenorm_update(vn_state, moments, beta)); neither depends on this optimiser step. */
int orl_ppo_apply_perm(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums,
                       const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                       float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id,
                       float* vn_state, const double* moments, double beta, void* stream);

/* ---- multi-GPU: one-shot small-message SUM all-reduce over xGMI peer memory ----------
 * The PPO path shards env lanes over the GPUs of one node (one process per GPU); the only data-path exchange is ONE
 * flat fp32 vector per optimiser step (both towers' raw gradient sums + denominators + logging sums, 38.9 KB at
 * configuration 2) - the generalisation of the reference's only live collective, the sum-all-reduce of
 * openrl/modules/networks/utils/distributed_utils.py:22-26, and of the DDP hook left empty at
 * openrl/algorithms/ppo.py:437-443.  At that size a ring is latency-bound, so every rank PUSHES its vector to every
 * peer's inbox (hipIpc-mapped device memory, one hop) as 8-byte {fp32, sequence tag} granules and sums the G
 * contributions in RANK ORDER: all ranks compute the bit-identical result, independent of arrival order.
 *
 * An orl_comm is the one object of this ABI that owns memory (its inbox, 2 x world x capacity granules, and the
 * mapped peer inboxes).  Set-up: every rank calls orl_comm_create (-> its 64-byte IPC handle), the caller exchanges
 * the handles out of band (e.g. torch.distributed.all_gather), every rank calls orl_comm_connect with all of them
 * ([world][64] bytes, rank order).  All ranks must issue the same sequence of collectives on a comm.  A peer that
 * does not arrive within 10 s sets an error word (orl_comm_error; it synchronises the stream) instead of hanging. */
typedef struct orl_comm orl_comm;
#define ORL_IPC_HANDLE_BYTES 64
int orl_comm_create(int rank, int world, int64_t capacity_floats, orl_comm** comm_out, unsigned char* handle_out);
int orl_comm_connect(orl_comm* comm, const unsigned char* all_handles);
int orl_comm_destroy(orl_comm* comm);
int orl_comm_error(orl_comm* comm, void* stream);
/* The same error word copied into a caller-owned int32 on the device, asynchronously on `stream` (no synchronisation):
 * the caller folds it into a read-back it does anyway (the train_info scalars), so a timeout in the middle of an update
 * surfaces at the end of that update instead of never. */
int orl_comm_error_copy(orl_comm* comm, int* err_out_dev, void* stream);
/* data[0..n) <- sum over ranks of data[0..n), in place, n <= capacity; no-op for world == 1. */
int orl_allreduce_small(orl_comm* comm, float* data, int n, void* stream);
/* The same collective fused into the optimiser step, so that a multi-GPU step is the same TWO launches as a
 * single-GPU one: orl_ppo_reduce_pair_comm = orl_ppo_reduce_pair whose workgroups also push every column sum they
 * produce to the peers (opens a collective); orl_ppo_apply_comm = orl_ppo_apply(_perm when next_idx != NULL) that sums
 * the G contributions in rank order while it stages the raw sums (closes it) and writes the global sums back to
 * `sums`.  The two calls must come in this order, once each per optimiser step. */
int orl_ppo_reduce_pair_comm(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy,
                             int n_blocks_critic, int width_critic, float* sums, void* stream);
int orl_ppo_apply_comm(orl_comm* comm, const orl_net_desc* pnet, const orl_net_desc* cnet, float* sums,
                       const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                       float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id,
                       float* vn_state, const double* moments, double beta, void* stream);

/* The optimiser step of an MLP-tower minibatch in ONE launch = orl_ppo_reduce_pair(_comm) + orl_ppo_apply(_perm / _comm)
 * with the same arguments and the same results (same summation orders, same Adam arithmetic): the reducing workgroups
 * take a ticket per tower, and the one that draws a tower's last ticket runs that tower's step (clip_grad_norm_ + Adam,
 * openrl/algorithms/ppo.py:132-164) behind a device-scope fence; the others share the next epoch's permutation job
 * (next_idx != NULL).  Measured on MI355X it does NOT pay: the device-scope release / acquire around the ticket (L2
 * write-back + invalidate across 8 XCDs) costs more than the kernel boundary it removes (15.7 us against 4.4 + 9.3 us);
 * the host mirror keeps the two-launch form as its default (cfg.amd_optim_step).  comm == NULL: single GPU; else the collective of the _comm pair, opened and
 * closed by this launch.  sync_ctr: 4 x uint32 of device memory, zero before the first call, owned by the calls that
 * share a stream (every launch leaves it zero). */
int orl_ppo_reduce_apply(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy,
                         int n_blocks_critic, int width_critic, float* sums, const orl_net_desc* pnet,
                         const orl_net_desc* cnet, const orl_ppo_hparams* hp, const orl_adam_state* padam,
                         const orl_adam_state* cadam, float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed,
                         uint64_t stream_id, float* vn_state, const double* moments, double beta, uint32_t* sync_ctr,
                         void* stream);

/* Round 6: the optimiser step of an MLP-tower minibatch in ONE launch, same arguments and same results as
 * orl_ppo_reduce_pair(_comm) + orl_ppo_apply(_perm / _comm) (bit-identical sums, same clip_grad_norm_ + Adam arithmetic,
 * openrl/algorithms/ppo.py:132-164).  Unlike orl_ppo_reduce_apply the two optimiser workgroups are DESIGNATED and launched
 * with the reducing ones: they fetch parameters and Adam moments while the column sums are formed, wait on their tower's
 * ticket word, and read the sums - published write-through, no L2 write-back / invalidate fence on either side - with
 * system-scope loads.  Only those two workgroups ever wait (bounded; a timeout writes NaN into the train_info slots), so no
 * co-residency of the grid is assumed.  sync_ctr: 4 x uint32 of device memory, zero before the first call. */
int orl_ppo_step(orl_comm* comm, const float* partials, int n_blocks_policy, int width_policy, int n_blocks_critic,
                 int width_critic, float* sums, const orl_net_desc* pnet, const orl_net_desc* cnet,
                 const orl_ppo_hparams* hp, const orl_adam_state* padam, const orl_adam_state* cadam,
                 float* train_info_accum, int64_t* next_idx, int64_t n, uint64_t seed, uint64_t stream_id, float* vn_state,
                 const double* moments, double beta, uint32_t* sync_ctr, void* stream);

/* ValueNorm.update (openrl/modules/utils/valuenorm.py:58-77) from reduced batch sums:
 * moments = {sum(x), sum(x^2), count} as doubles on the device; beta = 0.99999. */
int orl_valuenorm_update(float* vn_state, const double* moments, double beta, void* stream);
/* sum / sumsq / count of returns over minibatch rows (records column `ret_col`) -> moments[3]. */
int orl_minibatch_moments(const float* records, int rec_width, int ret_col, const int64_t* idx, int mb,
                          double* scratch, double* moments, void* stream);

/* ---- fused device-resident rollout ---------------------------------------
 * One launch performs `T` steps of {policy+value forward, sample, env.step, buffer insert} for a
 * device-resident batched env, i.e. OnPolicyDriver.actor_rollout (onpolicy_driver.py:154-203) with
 * act (:235-279) and add2buffer (:80-152) fused; env lanes never leave the GPU.
 */
#define ORL_ENV_SYNTH 0    /* fixed-step synthetic env: obs ~ N(0,1) keyed (seed, env, t), reward U(0,1) */
#define ORL_ENV_CARTPOLE 1 /* CartPole-v1 dynamics (gymnasium classic_control cartpole.py) */
#define ORL_ENV_TTT 2      /* tic-tac-toe vs a uniformly random opponent (orl_ttt_*): obs 18, Discrete(9), legal-move masks */
#define ORL_ENV_TTT_POOL 3 /* the same game, the opponent of env group g is policy g of a pool (orl_rollout_args.opp_*) */

typedef struct orl_rollout_args {
  orl_buffer_ptrs buf;
  float* value_preds;      /* [T+1, N, A, 1] */
  float* actions;          /* [T, N, A, a]   */
  float* action_log_probs; /* [T, N, A, a]   */
  float* env_state;        /* [N, env_state_width] persistent env state */
  float* ep_stats;         /* [N, 4]: running episode return, length, sum of finished returns, finished count */
  int32_t env_kind;
  int32_t episode_limit;   /* synthetic: fixed episode length; cartpole: 500 */
  uint64_t env_seed;
  uint64_t act_seed;
  uint64_t rng_step0;      /* global step counter at the first step of this rollout */
  /* ORL_ENV_TTT_POOL only: a pool of opponent policies with the policy tower's architecture (orl_act_step_grouped) */
  const float* opp_thetas; /* parameters of policy g at opp_thetas + g*opp_theta_stride */
  int64_t opp_theta_stride;
  int32_t opp_group_rows;  /* envs [g*opp_group_rows, ...) play policy g; a multiple of 16 */
  int32_t opp_reserved;    /* ORL_ENV_SYNTH / ORL_ENV_CARTPOLE / ORL_ENV_TTT: 0 = the round-6 chain rollout (policy-only step chain, the
                            * critic on background waves of the same launch), 1 = the round-5 kernel (both towers in the
                            * step loop).  Ignored by the tic-tac-toe POOL envs (always the round-5 kernel). */
  uint64_t opp_seed;       /* Philox seed of the opponents' sampling; counter = (env, opp_rng_step0 + t) */
  uint64_t opp_rng_step0;
  const int32_t* opp_index; /* optional [N]: env n plays pool policy opp_index[n] instead of n / opp_group_rows; unless
                             * opp_per_reset is set the fused kernel needs it uniform over each 16-env tile
                             * (orl_opponent_sample, per_tile) */
  /* opp_per_reset != 0 (OpponentPoolWrapper.reset semantics, opponent_pool_wrapper.py:37-66, inside the fused kernel):
   * every env names its own slot in opp_index, which is READ AND WRITTEN - a finished game draws its next opponent
   * in-kernel exactly as orl_opponent_sample(per_tile = 0, draw_id = opp_draw_id0 + t) would.  All opp_n_policies
   * (<= 4) snapshot images stay resident in LDS; the critic then runs as one batched launch after the step loop. */
  int32_t opp_per_reset;
  int32_t opp_n_policies;   /* pool slots (images at opp_thetas + k * opp_theta_stride) */
  int32_t opp_n_filled;     /* slots a RandomOpponent draw may return: [0, opp_n_filled) */
  int32_t opp_last_slot;    /* LastOpponent's slot */
  int32_t opp_strategy;     /* 0 RandomOpponent, 1 LastOpponent (orl_opponent_sample) */
  int32_t opp_pad;
  uint64_t opp_sample_seed; /* orl_opponent_sample's seed */
  uint64_t opp_draw_id0;    /* draw id of the rollout's first step */
} orl_rollout_args;

/* 1 when the loaded library was built with -DORL_BUILD_EXPERIMENTS=1 (the comparison kernels that lost their A/B are present:
 * hparams.reserved & 4 / & 8 of orl_ppo_fwd_bwd, & 8 of orl_rnn_ppo_fwd_bwd, orl_ppo_reduce_apply), 0 for the shipped build. */
int orl_build_experiments(void);
/* How the MLP towers' 64-wide GEMMs (orl_ppo_fwd_bwd, the chain rollout's critic) form their fp32 products on the 16-bit MFMA:
 * 2 = two-term fp16 splits, 3 products, operands scaled by exact powers of two (the shipped build, round 6);
 * 3 = three-term bf16 splits, 6 of 9 products (rounds 3 - 5; ORL_BUILD_DEFS=-DORL_TOWER_F16=0).  Both at fp32 accuracy.
 * (The recurrent towers' data_chunk_length == 2 row kernel of orl_rnn_ppo_fwd_bwd forms its 64-wide products the same two-term way
 * over fp16 images of its seven matrices - ORL_RNN_L2_H2, on by default, independent of this value; every other recurrent kernel
 * uses the fp32 MFMA.) */
int orl_tower_split_terms(void);

int orl_env_state_width(int env_kind);
int orl_env_reset(int env_kind, float* env_state, float* ep_stats, float* obs0, int N, int obs_dim,
                  uint64_t env_seed, int episode_limit, void* stream);
/* One env.step of a device env outside the fused rollout (evaluation loops, the stepwise driver; the
 * VecEnv.step contract of openrl/envs/vec_env/base_venv.py): auto-reset semantics, obs [N, obs_dim],
 * rewards [N], dones uint8 [N].  `global_step` is the env's step counter (keys the synthetic stream). */
int orl_env_step(int env_kind, float* env_state, float* ep_stats, const float* actions, int action_width,
                 float* obs, float* rewards, uint8_t* dones, int N, int obs_dim, uint64_t env_seed,
                 int episode_limit, uint64_t global_step, void* stream);
/* The same step with the counter split in two: effective step = global_step + *global_step_dev (mod 2^64).  A stepwise
 * rollout captured in a hipGraph freezes the host part; the device part (an int64 the graph itself advances) carries the
 * progress from replay to replay.  global_step_dev == NULL is orl_env_step. */
int orl_env_step_dev(int env_kind, float* env_state, float* ep_stats, const float* actions, int action_width,
                     float* obs, float* rewards, uint8_t* dones, int N, int obs_dim, uint64_t env_seed,
                     int episode_limit, uint64_t global_step, const int64_t* global_step_dev, void* stream);

/* V(obs) of `rows` stored observations [rows, obs_dim] in one persistent launch (PPOModule.get_values,
 * ppo_module.py:140-147, for a whole rollout at once); the same per-row arithmetic as orl_act_step's critic. */
int orl_critic_values(const orl_net_desc* cnet, const float* ctheta, const float* critic_obs, int64_t rows,
                      float* values, void* stream);

/* next_value [N] (optional): critic value of the observation in slot T, i.e. the bootstrap value
 * OnPolicyDriver.compute_returns feeds ReplayData.compute_returns (onpolicy_driver.py:205-233). */
int orl_rollout_fused(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet,
                      const float* ctheta, const orl_rollout_args* args, float* next_value, void* stream);

/* ---- device-resident MPE simple_spread (BASELINE config 4's env) ------------------
 * 3 agents, 3 landmarks, Discrete(5) actions: World.step physics (openrl/envs/mpe/core.py:216-323), rewards and
 * observations of scenarios/simple_spread.py:84-125, shared reward and done-at-world_length of
 * multiagent_env.py:167-260, auto-reset of envs/vec_env/sync_venv.py:217-222.  One env = 3 buffer lanes.
 *   env_state [N, orl_mpe_state_width()]; actions [N,3] (action index as float32, the layout orl_*act_step writes);
 *   obs_policy [N,3,18]; obs_critic [N,3,54] (all agents' obs concatenated, may be NULL); rewards [N,3]; dones
 *   uint8 [N,3]; ep_stats [N,4] as for the other device envs (may be NULL).  Reset positions come from Philox keyed
 *   (env_seed, env, episode) instead of numpy's PCG64 (documented deviation; physics is generator-free). */
int orl_mpe_state_width(void);
int orl_mpe_reset(float* env_state, float* ep_stats, float* obs_policy, float* obs_critic, int N, uint64_t env_seed,
                  void* stream);
int orl_mpe_step(float* env_state, float* ep_stats, const float* actions, float* obs_policy, float* obs_critic,
                 float* rewards, uint8_t* dones, int N, uint64_t env_seed, int world_length, void* stream);

/* ---- device-resident tic-tac-toe vs a uniformly random opponent (BASELINE config 5's env) ---------------------
 * examples/selfplay: PettingZoo tictactoe_v3 behind RandomOpponentWrapper (selfplay/wrappers/
 * base_multiplayer_wrapper.py:85-150, random_opponent_wrapper.py:27-43).  One env = one game; a step is the agent's
 * move plus the opponent's reply; rewards +1 / -1 / 0 (illegal move: -1, game over); obs [N, 18], action masks
 * [N, 9] (1 = empty cell), rewards [N], dones uint8 [N]; finished games restart in the same step (auto-reset). */
int orl_ttt_state_width(void);
int orl_ttt_reset(float* env_state, float* ep_stats, float* obs, float* action_masks, int N, uint64_t env_seed,
                  void* stream);
int orl_ttt_step(float* env_state, float* ep_stats, const float* actions, float* obs, float* action_masks,
                 float* rewards, uint8_t* dones, int N, uint64_t env_seed, void* stream);
/* The same step in two launches, for an opponent that is a POLICY (self-play; OpponentPoolWrapper of the reference
 * plays earlier checkpoints): orl_ttt_agent_move applies the agent's moves and writes, for every game still open, the
 * board from the opponent's side (opp_obs [N, 18], opp_masks [N, 9]); the caller samples opp_actions [N] from the
 * opponent's policy (orl_act_step); orl_ttt_opponent_move applies them, settles rewards / dones, auto-resets and
 * writes the agent's next observation / mask.  An opening move of the opponent after a reset stays uniform. */
int orl_ttt_agent_move(float* env_state, const float* actions, float* opp_obs, float* opp_masks, float* rewards,
                       uint8_t* dones, int N, void* stream);
int orl_ttt_opponent_move(float* env_state, float* ep_stats, const float* opp_actions, float* obs, float* action_masks,
                          float* rewards, uint8_t* dones, int N, uint64_t env_seed, void* stream);

/* ---- recurrent (GRU) towers: use_recurrent_policy -------------------------
 * Tower = MLPBase (as above) -> RNNLayer = one-layer nn.GRU(H,H) + LayerNorm(H) (openrl/modules/networks/utils/
 * rnn.py:5-99, recurrent_N = 1) -> head.  The orl_net_desc fields keep their meaning; `theta` is the reference's
 * parameter order for that network (policy_network.py:60-118, value_network.py:60-111):
 *   W1 b1 g1 be1 W2 b2 g2 be2 | weight_ih_l0[3H*H] weight_hh_l0[3H*H] bias_ih_l0[3H] bias_hh_l0[3H] (gates r,z,n)
 *   | g3[H] be3[H] (rnn.norm) | W3 b3 (logstd).
 * Step semantics (rnn.py:39-49): h' = GRU(base(obs), h * mask); features = LayerNorm(h').
 */
int orl_rnn_param_count(const orl_net_desc* net);
int orl_rnn_raw_grad_count(const orl_net_desc* net);

/* PPOModule.get_actions with recurrent networks (ppo_module.py:102-138): like orl_act_step plus the hidden
 * states h_*_in [B,H] (multiplied by masks [B] first) and the new states h_*_out [B,H] (may alias the inputs).
 * NULL ptheta = value-only call (get_values, :140-147); NULL ctheta = policy only. */
int orl_rnn_act_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                     const float* policy_obs, const float* critic_obs, const float* h_policy_in,
                     const float* h_critic_in, const float* masks, const float* action_masks, int B,
                     int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                     const float* forced_u, float* values, float* actions, float* logp, float* h_policy_out,
                     float* h_critic_out, void* stream);

/* One step of PPOModule.evaluate_actions with recurrent networks (ppo_module.py:149-193, rnn.py:39-99): like
 * orl_rnn_act_step, but the GIVEN actions [B, a] are evaluated - log-probs [B, a], per-row entropy [B] - instead of
 * sampled.  A sequence of L steps is L calls that hand h_*_out to the next call's h_*_in. */
int orl_rnn_eval_step(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                      const float* policy_obs, const float* critic_obs, const float* h_policy_in,
                      const float* h_critic_in, const float* masks, const float* action_masks, const float* actions,
                      int B, float* values, float* logp, float* entropy, float* h_policy_out, float* h_critic_out,
                      void* stream);

/* ReplayData.recurrent_generator's index arithmetic (buffers/replay_data.py:1062-1258): chunk c covers the rows
 * c*L .. c*L+L-1 of the [lane][t]-ordered flat batch (lane = n*A+a; chunks may straddle lanes when T % L != 0).
 * rows[l*n_chunks + i] = record row (t*lanes + lane) of step l of chunk chunk_idx[i] (NULL = identity) - the same
 * index addresses masks [T+1,lanes] and, scaled by H, the stored rnn states [T+1,lanes,H]. */
int orl_rnn_chunk_rows(const int64_t* chunk_idx, int n_chunks, int L, int T, int lanes, int64_t* rows, void* stream);

/* recurrent_generator_v3 (buffers/replay_data.py:425-551, use_joint_action_loss): chunk c = positions c*L .. c*L+L-1
 * of the (n*T + t)-ordered batch with the agent axis kept; rows[l*(n_chunks*A) + i*A + a] = record row of agent a at
 * step l of chunk chunk_idx[i] (agent0_only: rows[l*n_chunks + i], a = 0 - the critic's sequences). */
int orl_rnn_chunk_rows_v3(const int64_t* chunk_idx, int n_chunks, int L, int T, int n_envs, int n_agents, int agent0_only,
                          int64_t* rows, void* stream);
/* Joint-action loss (JRPO, algorithms/ppo.py:254-300): given the CURRENT policy's log-probs logp_new [L][n_chunks*A][a]
 * of the minibatch rows (orl_rnn_eval_step), write to records_out (a full copy of `records`) the old log-probs shifted
 * so that exp(logp - old') is the JOINT ratio over agents and action dims, adv' = A * agent 0's advantage and
 * active' = agent 0's active mask for every agent row of a (step, chunk); the per-row PPO loss of
 * orl_rnn_ppo_fwd_bwd on records_out then has exactly the joint-action gradient. */
int orl_rnn_jrpo_records(const float* records, float* records_out, int rec_width, int Dp, int Dc, int a_w,
                         const int64_t* rows, int n_chunks, int L, int n_agents, const float* logp_new, void* stream);

typedef struct orl_rnn_batch {
  const float* records;  /* packed update records (orl_adv_normalize_pack), row = t*lanes + lane */
  const int64_t* rows;   /* [L][n_chunks] from orl_rnn_chunk_rows */
  const float* masks;    /* [T+1, lanes] */
  const float* h_policy; /* [T+1, lanes, H] rnn_states          */
  const float* h_critic; /* [T+1, lanes, H] rnn_states_critic   */
  int32_t rec_width, n_chunks, L, n_chunks_critic;
  const int64_t* rows_critic; /* optional [L][n_chunks_critic]: the critic tower's own sequences (joint-action loss:
                               * agent 0 of every chunk); NULL = the policy's rows */
} orl_rnn_batch;

/* float32 scratch the update needs (activation tapes + per-workgroup partials), in floats */
int64_t orl_rnn_workspace_floats(const orl_net_desc* pnet, const orl_net_desc* cnet, int n_chunks, int L);

/* PPOAlgorithm.prepare_loss + 2x backward (ppo.py:98-124, 238-361) on one recurrent minibatch: per tower a
 * row kernel (forward over the L steps, loss, back-propagation through time; activation gradients go to a tape)
 * and a weight-gradient GEMM kernel over the tape, then the deterministic reduction into
 *   sums[raw_p + ORL_N_STATS + raw_c + ORL_N_STATS]   (raw_* = orl_rnn_raw_grad_count)
 * - the vector a multi-GPU run all-reduces. */
int orl_rnn_ppo_fwd_bwd(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                        const orl_rnn_batch* batch, const float* vn_state, const orl_ppo_hparams* hp,
                        float* workspace, float* sums, void* stream);

/* orl_ppo_apply for recurrent towers; `scratch` holds 512 floats. */
int orl_rnn_ppo_apply(const orl_net_desc* pnet, const orl_net_desc* cnet, const float* sums, const orl_ppo_hparams* hp,
                      const orl_adam_state* padam, const orl_adam_state* cadam, float* train_info_accum,
                      float* scratch, void* stream);

/* OnPolicyDriver.actor_rollout (openrl/drivers/onpolicy_driver.py:154-233) of a recurrent policy on a device-resident
 * multi-agent env, fused: policy + env workgroups step all T = buf.T steps (hidden states in registers, worlds stepped
 * in-kernel like orl_mpe_step, every per-step buffer field written from the kernel incl. masks[t+1] and
 * rnn_states[dones_env] = 0 of add2buffer, :80-152), critic workgroups sweep share_obs and fill
 * value_preds[0..T), rnn_states_critic[1..T] and `next_value` (the bootstrap value compute_returns needs, :205-233) -
 * in the same launch when `sync_flags` is given, as a second launch otherwise.  On ORL_ENV_MPE_SPREAD share_obs of slots 1..T
 * (the concatenation of a world's three observations, envs/mpe/environment.py) is read by the critic workgroups from the world's
 * three contiguous rows of policy_obs and WRITTEN by them to critic_obs; slot 0 of critic_obs is the caller's.
 * Reads slot 0 of obs / masks / rnn_states* (ReplayData.after_update / init_buffer put them there).  Same per-row
 * arithmetic and Philox counters (act_seed, row, rng_step0 + t [+ *rng_step_dev]) as T x {orl_rnn_act_step,
 * orl_mpe_step, orl_buffer_insert_rnn}.  Built for env_kind ORL_ENV_MPE_SPREAD (3 agents, obs 18 / 54, Discrete(5)). */
#define ORL_ENV_MPE_SPREAD 4
typedef struct orl_rnn_rollout_args {
  orl_buffer_ptrs buf;
  float* value_preds;        /* [T+1, N, A, 1] */
  float* actions;            /* [T, N, A, 1]   */
  float* action_log_probs;   /* [T, N, A, 1]   */
  float* rnn_states;         /* [T+1, N, A, H] */
  float* rnn_states_critic;  /* [T+1, N, A, H] */
  float* env_state;          /* [N, orl_mpe_state_width()] */
  float* ep_stats;           /* [N, 4] or NULL */
  float* obs_policy_out;     /* optional [N, A, Dp]: the env's own observation arrays, left as after the last step */
  float* obs_critic_out;     /* optional [N, A, Dc] */
  float* next_value;         /* optional [N*A] */
  int32_t env_kind, world_length, deterministic, reserved;
  uint64_t env_seed, act_seed, rng_step0;
  const uint64_t* rng_step_dev; /* optional device-side addend of rng_step0 (see orl_act_step) */
  int32_t* sync_flags;       /* optional [ceil(N/16) + 1] int32 scratch: with it the critic workgroups run in the SAME
                              * launch, one step behind their policy workgroups (per-group step counters, cleared by every
                              * call; the last word is set to 1 if a critic's bounded wait - or a bounded wait between the
                              * four waves that share a 16-row tile - ever timed out - STICKY: the caller zeroes it once and it
                              * is never cleared here); NULL = two launches */
  uint64_t env_step0;        /* ORL_ENV_SYNTH / ORL_ENV_CARTPOLE (single agent, A == 1, one shared observation array; then
                              * world_length = the episode limit, env_state as orl_env_reset leaves it, sync_flags unused):
                              * the env's global step at the first rollout step */
} orl_rnn_rollout_args;
int orl_rnn_rollout_fused(const orl_net_desc* pnet, const float* ptheta, const orl_net_desc* cnet, const float* ctheta,
                          const orl_rnn_rollout_args* args, void* stream);

/* ==== general tower path (csrc/orl_gen.hip) ==========================================================================
 * The fused kernels above are built for the reference's DEFAULT tower (hidden_size 64, layer_N 1, ReLU, no feature
 * LayerNorm) - what every BASELINE.json configuration runs.  Everything else MLPBase / MLPLayer
 * (openrl/modules/networks/utils/mlp.py:8-46,100-180), PolicyValueNetwork (use_share_model,
 * openrl/modules/networks/policy_value_network.py:34-230) and ACTLayer's MultiDiscrete branch
 * (openrl/modules/networks/utils/act.py:26-34,60-72,136-151) can express runs layer by layer through these entry
 * points, activations in caller-owned device buffers; the layer loop is host code (modules/generic_net.py), where the
 * reference has it (nn.Module.forward). */
#define ORL_ACT_NONE (-1)
#define ORL_ACT_TANH 0 /* = cfg.activation_id (mlp.py:13) */
#define ORL_ACT_RELU 1
#define ORL_ACT_LEAKY_RELU 2
#define ORL_ACT_ELU 3
#define ORL_HEAD_MULTI_DISCRETE 3 /* MultiDiscrete(nvec): n_heads Categoricals over one concatenated logits row */
/* Tuple(Box(cd), Discrete(n)) - the reference's "mixed" ACTLayer branch (act.py:33-63, 126-147): a DiagGaussian over cd
 * dims and a Categorical over n classes on the same features; logits row = [means (cd) | class logits (n)], nvec = {cd, n};
 * stored actions = cd floats + the class index, ONE joint log-prob (the sum over both parts) replicated over the cd + 1
 * stored columns, dist_entropy = 0.0025 * Gaussian + 0.01 * Categorical (both hard-coded there). */
#define ORL_HEAD_MIXED 4
#define ORL_MAX_HEADS 8
typedef struct orl_head_desc {
  int32_t kind;    /* ORL_HEAD_CATEGORICAL / _GAUSSIAN / _MULTI_DISCRETE / _MIXED */
  int32_t n_out;   /* logits per row (MultiDiscrete: sum of nvec) */
  int32_t n_heads; /* MultiDiscrete: components, else 1 */
  int32_t nvec[ORL_MAX_HEADS];
} orl_head_desc;

/* C[M,N] = sum_k A(m,k) * B(k,n) in fp32 on MFMA; element (m,k) of A sits at A + m*sam + k*sak, (k,n) of B at
 * B + k*sbk + n*sbn, so the same kernel is x W^T (nn.Linear forward), dz W (input gradient) and dz^T x (weight
 * gradient: K = batch rows).  n_split > 1 splits K into that many slices whose partial products go to
 * partials[n_split][M*N] and are summed in slice order (deterministic). */
int orl_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc,
             int M, int N, int K, int n_split, float* partials, void* stream);
/* One layer of MLPLayer below its Linear: a = act(z + bias); with gamma/beta also LayerNorm (eps 1e-5):
 * xhat = (a - mean) * rstd, y = xhat * gamma + beta; without them y = a.  z, a, xhat, y are [B, H] (H <= 512), rstd [B];
 * every output pointer may be NULL.  Also used with z = observations, act NONE, bias NULL for MLPBase.feature_norm. */
int orl_row_fwd(const float* z, const float* bias, int act, const float* gamma, const float* beta, int B, int H,
                float* a_out, float* xhat_out, float* rstd_out, float* y_out, void* stream);
/* Its backward: dy [B,H] -> dz [B,H] (gradient at the Linear's output; may be NULL) and per-workgroup partial rows
 * col_partials[n_blocks][3H] = [d gamma | d beta | d bias] to be column-summed with orl_ppo_reduce. */
int orl_row_bwd(const float* dy, const float* gamma, const float* xhat, const float* rstd, const float* a, int act, int B,
                int H, float* dz_out, float* col_partials, int max_blocks, int* n_blocks_out, void* stream);
/* Fused layer of MLPLayer (mlp.py:8-46), nn.Sequential(Linear, act, LayerNorm) in ONE launch: y = LN(act(x W^T + b))
 * for x [B, n_in], W [n_out, n_in] (nn.Linear layout), n_out <= 512.  a_out [B, n_out] (post-activation, pre-LayerNorm)
 * and stats_out [B, 2] = (mean, rstd) per row are what orl_gen_layer_bwd needs; either may be NULL (rollouts).  Without
 * gamma / beta (action / value heads) y = act(x W^T + b) and stats_out is not written. */
int orl_gen_layer_fwd(const float* x, int B, int n_in, const float* W, const float* bias, int act, const float* gamma,
                      const float* beta, int n_out, float* a_out, float* stats_out, float* y_out, void* stream);
/* The whole tower of a rollout step in ONE launch (no activations stored): optional feature LayerNorm over the
 * observation, n_layers x nn.Sequential(Linear, act, LayerNorm) with n_in(k) == n_out(k-1), widths multiples of 4 and
 * <= 256, then n_heads (0, 1 or 2) plain Linear heads on the trunk's features: entries layer[n_layers ..].  head_out0 /
 * head_out1 are [B, n_out of that head]; feats_out (or NULL) receives the trunk's features [B, n_out(n_layers-1)] - what
 * a recurrent cell between trunk and head consumes.  All pointers are device pointers (typically into one flat
 * parameter vector). */
#define ORL_GEN_MLP_MAX_LAYERS 14
typedef struct orl_gen_mlp_layer {
  const float* W;      /* [n_out, n_in] */
  const float* bias;   /* [n_out] or NULL */
  const float* gamma;  /* LayerNorm weight [n_out]; NULL for a head */
  const float* beta;
  int32_t n_in, n_out, act, reserved;
} orl_gen_mlp_layer;
typedef struct orl_gen_mlp_desc {
  int32_t n_layers, n_heads;
  const float* fn_gamma; /* MLPBase.feature_norm over the n_in(0) observation columns, or NULL */
  const float* fn_beta;
  orl_gen_mlp_layer layer[ORL_GEN_MLP_MAX_LAYERS];
} orl_gen_mlp_desc;
int orl_gen_mlp_fwd(const orl_gen_mlp_desc* desc, const float* x, int B, float* head_out0, float* head_out1,
                    float* feats_out, void* stream);
/* One rollout step of a policy / critic pair in ONE launch (OnPolicyDriver.act -> PPOModule.get_actions:
 * openrl/modules/ppo_module.py:118-150 -> PolicyNetwork.forward + ValueNetwork.forward): the policy tower as
 * orl_gen_mlp_fwd runs it, ACTLayer.forward on its first head's logits exactly as orl_gen_sample does it (same
 * arithmetic, same Philox counters: the two routes agree bit for bit), and either the critic tower on critic_obs
 * (critic != NULL: grid.y = 2) or the shared network's second head (policy->n_heads == 2, critic == NULL).
 * logits_out [B, head->n_out] may be NULL; values [B] is NULL exactly when there is no value head. */
int orl_gen_act(const orl_gen_mlp_desc* policy, const float* obs, const orl_gen_mlp_desc* critic, const float* critic_obs,
                int B, float* logits_out, float* values, const orl_head_desc* head, const float* logstd,
                const float* action_masks, int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step,
                const uint64_t* rng_step_dev, const float* forced_u, int a_w, float* actions, float* logp, void* stream);
/* The fused rollout of the general tower path: all buf->T steps of {policy tower (+ a shared network's value head, policy
 * descriptor with two heads), ACTLayer sampling, env.step of a device-resident single-agent env (ORL_ENV_SYNTH /
 * ORL_ENV_CARTPOLE), ReplayData.insert} in ONE launch - openrl/drivers/onpolicy_driver.py:154-203 for towers outside the
 * default one.  Reads slot 0 of the buffer, fills slots 1..T (observations, masks), rewards, actions / log-probs [T][N][a_w]
 * and, with a value head, value_preds[0..T).  A separate critic is evaluated by the caller afterwards in one
 * orl_gen_mlp_fwd launch over all (T + 1) N rows.  Philox counters as orl_gen_act at rng_step0 + t; the env advances from
 * its global step env_step0. */
int orl_gen_rollout_fused(const orl_gen_mlp_desc* policy, const orl_head_desc* head, const float* logstd,
                          const orl_buffer_ptrs* buf, float* value_preds, float* actions, float* action_log_probs,
                          float* env_state, float* ep_stats, int env_kind, int episode_limit, uint64_t env_seed,
                          uint64_t env_step0, uint64_t act_seed, uint64_t rng_step0, int a_w, void* stream);
/* orl_gen_layer_fwd's backward in one launch: dy [B, n_out] -> dz_out [B, n_out] (gradient at the Linear's output; may be NULL),
 * col_partials[n_blocks][3 n_out] = [d gamma | d beta | d bias] per workgroup (sum them with orl_gen_colsum), and - for a
 * square layer, when dx_out != NULL - the input gradient dx_out [B, n_in] = dz W of the same row tile without re-reading
 * dz.  a / stats as written by orl_gen_layer_fwd (stats unused without gamma, a unused without gamma and activation). */
int orl_gen_layer_bwd(const float* dy, const float* a, const float* stats, const float* gamma, int act, int B, int n_out,
                      const float* W, int n_in, float* dz_out, float* dx_out, float* col_partials, int max_blocks,
                      int* n_blocks_out, void* stream);
/* dW [n_out, n_in] = dz [B, n_out]^T x [B, n_in]: persistent split-K over the batch rows (slice partial products in
 * partials, summed in slice order: deterministic).  partials_floats >= n_out * n_in; more lets more workgroups run. */
int orl_gen_wgrad(const float* dz, const float* x, int B, int n_out, int n_in, float* dW, float* partials,
                  int64_t partials_floats, void* stream);
/* Fixed-order column sums of partials[n_rows][width]; columns [0, w0) go to dst0, [w0, w0+w1) to dst1, the rest to
 * dst2 (w0 + w1 + w2 == width; a NULL destination drops its segment). */
int orl_gen_colsum(const float* partials, int n_rows, int width, float* dst0, int w0, float* dst1, int w1, float* dst2,
                   int w2, void* stream);
/* out[i, :] = records[idx[i] (or i), col0 : col0+width] */
int orl_gather_cols(const float* records, int rec_width, int col0, int width, const int64_t* idx, int mb, float* out,
                    void* stream);
/* den[0] = sum of the active column over the minibatch rows, den[1] = rows: the masked-mean denominators of
 * PPOAlgorithm.prepare_loss (ppo.py:319-361); a multi-GPU run all-reduces these two floats.  `scratch`: 257 floats
 * the caller zeroes ONCE (slice sums + a completion counter that every call leaves at zero again). */
int orl_gen_denoms(const float* records, int rec_width, int Dp, int Dc, int a_w, const int64_t* idx, int mb, float* den,
                   float* scratch, void* stream);
/* Policy part of prepare_loss + ACTLayer.evaluate_actions for rows idx[0..mb) given the head's logits [mb, n_out]:
 * training (dlogits != NULL): dlogits = d(policy_loss - entropy_coef * dist_entropy) / d logits, ALREADY divided by
 * the denominators in den; partials[n_blocks][20] = {policy-loss sum, entropy sum, ratio sum, -, dlogstd[16]}.
 * evaluation (dlogits == NULL): logp_out [mb, a_w] and ent_out [mb] only. */
int orl_gen_policy_loss(const orl_head_desc* head, const float* logits, const float* logstd, const float* records,
                        int rec_width, int Dp, int Dc, int a_w, int K, const int64_t* idx, int mb, const float* den,
                        const orl_ppo_hparams* hp, float* dlogits, float* partials, int max_blocks, int* n_blocks_out,
                        float* logp_out, float* ent_out, void* stream);
/* cal_value_loss (ppo.py:178-220): dvalues [mb] (already * value_loss_coef / denominator), partials[n_blocks][1]. */
int orl_gen_value_loss(const float* values, const float* records, int rec_width, int Dp, int Dc, int a_w, int K,
                       const int64_t* idx, int mb, const float* vn_state, const float* den, const orl_ppo_hparams* hp,
                       float* dvalues, float* partials, int max_blocks, int* n_blocks_out, void* stream);
/* ACTLayer.forward from logits: actions / log-probs [B, a_w]; Philox counters as orl_act_step (MultiDiscrete component
 * h takes word h of the stream). */
int orl_gen_sample(const orl_head_desc* head, const float* logits, const float* logstd, const float* action_masks, int B,
                   int deterministic, uint64_t seed, uint64_t row0, uint64_t rng_step, const uint64_t* rng_step_dev,
                   const float* forced_u, int a_w, float* actions, float* logp, void* stream);
/* clip_grad_norm_ (n_clips = 2: twice, as ppo.py:127-145 does for the shared model) + torch.optim.Adam on one flat
 * vector; train_info_accum[slot_first] += norm before the first clip, [slot_second] += norm before the second
 * (slots < 0 are skipped); scratch holds 256 floats. */
int orl_gen_adam(const orl_adam_state* adam, int64_t n, float max_grad_norm, int use_max_grad_norm, int n_clips,
                 float* scratch, float* train_info_accum, int slot_first, int slot_second, void* stream);
/* dst[0..n) += src[0..n): gradient accumulation when a shared trunk is back-propagated twice (policy and critic
 * observations differ) or two heads feed one feature gradient. */
int orl_vec_add(float* dst, const float* src, int64_t n, void* stream);
/* y [B, N] = x [B, K] W [K, N] (all row-major, fp32 MFMA): an input gradient dz W with W in nn.Linear's [out, in] layout,
 * through the tiled forward kernel (coalesced float4 loads of both operands, 128 output columns per launch). */
int orl_gen_matmul(const float* x, int B, int K, const float* W, int N, float* y, void* stream);
/* dst[c] = sum over the n_rows rows of x[r][c] (a bias gradient from a tall [rows, width] gradient matrix), in a fixed
 * order: row slabs into `partials` (>= width floats; up to 512 * width are used), then orl_gen_colsum. */
int orl_gen_colsum_rows(const float* x, int n_rows, int width, float* dst, float* partials, int64_t partials_floats,
                        void* stream);
/* One step of torch.nn.GRU (one layer; RNNLayer, networks/utils/rnn.py:28-99) after its two projections
 * gi = x W_ih^T + b_ih and gh = h_in W_hh^T + b_hh ([N, 3H], column blocks r | z | n): h_out = (1 - z) n + z h_in.
 * save [N, 4H] = (r, z, n, gh_n) feeds orl_gen_gru_gate_bwd (may be NULL); h_in_next = h_out * mask_next[row] is the
 * next step's masked input (both NULL or both given). */
int orl_gen_gru_gate_fwd(const float* gi, const float* gh, const float* h_in, const float* mask_next, int N, int H,
                         float* h_out, float* h_in_next, float* save, void* stream);
/* Its backward: dh [N, H] at h_out -> dgi, dgh [N, 3H] at the projections' outputs and dh_in = z * dh, the direct path
 * to h_in (the path through gh is dgh W_hh, a GEMM). */
int orl_gen_gru_gate_bwd(const float* dh, const float* save, const float* h_in, int N, int H, float* dgi, float* dgh,
                         float* dh_in, void* stream);
/* One step of torch.nn.LSTM (one layer; rnn_type lstm of RNNLayer) after its projections gi, gh [N, 4H] (blocks i | f | g | o):
 * c_out = f c_in + i g, h_out = o tanh(c_out).  save [N, 5H] = (i, f, g, o, tanh(c_out)) or NULL; h_in_next / c_in_next =
 * the next step's masked inputs (all three of them and mask_next, or none). */
int orl_gen_lstm_gate_fwd(const float* gi, const float* gh, const float* c_in, const float* mask_next, int N, int H,
                          float* h_out, float* c_out, float* h_in_next, float* c_in_next, float* save, void* stream);
/* Its backward: dh, dc (NULL = 0) at h_out / c_out -> dgates [N, 4H] (the gradient at both projections' outputs) and
 * dc_in, the gradient at c_in. */
int orl_gen_lstm_gate_bwd(const float* dh, const float* dc, const float* save, const float* c_in, int N, int H,
                          float* dgates, float* dc_in, void* stream);
/* out[row, :] = (a[row, :] + b[row, :]) * row_scale[row] + add[row, :]; b, row_scale, add may be NULL.  Masks on hidden
 * states and the carry of back-propagation through time. */
int orl_gen_row_affine(const float* a, const float* b, const float* row_scale, const float* add, int N, int H, float* out,
                       void* stream);
/* train_info_accum {value_loss, policy_loss, dist_entropy, -, -, ratio} += the reduced loss sums / denominators. */
int orl_gen_info(const float* policy_sums, const float* value_sums, const float* den, const orl_ppo_hparams* hp,
                 float entropy_div, float ratio_div, float* train_info_accum, void* stream);

/* ---- cross-layer fused general towers (csrc/orl_gen_tower.{h,hip}) ----------------------------------------------------
 * MLPBase.forward (openrl/modules/networks/utils/mlp.py:8-48, 100-180: optional feature LayerNorm, fc1, layer_N - 1 fc2
 * clones, fc3 - each nn.Sequential(Linear, [activation], LayerNorm)) + the Linear head(s) on its features
 * (ACTLayer's action_out / v_out: networks/policy_network.py:130-162, value_network.py:113-136,
 * policy_value_network.py:34-110), forward and backward, for hidden_size 64 / 128 with the activations kept on chip:
 * what orl_gen_layer_fwd / _bwd / orl_gen_wgrad do layer by layer through HBM.  The descriptor points into one flat
 * parameter vector (offsets in floats); layer 0 is [H, D], the others [H, H]; act[l] = ORL_ACT_* of layer l. */
#define ORL_GT_MAX_LAYERS 4
typedef struct orl_gt_desc {
  const float* theta;
  int32_t D, H, n_layers, n_heads;
  int32_t o_fn_g, o_fn_be; /* MLPBase.feature_norm weight / bias, -1 without feature normalisation */
  int32_t oW[ORL_GT_MAX_LAYERS], ob[ORL_GT_MAX_LAYERS], og[ORL_GT_MAX_LAYERS], obe[ORL_GT_MAX_LAYERS], act[ORL_GT_MAX_LAYERS];
  int32_t head_oW[2], head_ob[2], head_n[2]; /* head h: weight [head_n, H], bias [head_n]; all heads together <= 16 outputs */
} orl_gt_desc;
/* 1 when the fused kernels take this tower (hidden_size 64 or 128 with 2..4 layers, D <= 64, the weights and
 * the exchange slab within 160 KiB of LDS), else 0 (orl_last_error says why). */
int orl_gt_supported(const orl_gt_desc* d);
/* Sizes (floats) of the image orl_gt_prep writes and of the raw gradient-sum vector; -1 when unsupported. */
int64_t orl_gt_image_floats(const orl_gt_desc* d);
int64_t orl_gt_raw_floats(const orl_gt_desc* d);
/* theta -> image (LayerNorm affines folded into the following Linear, the H x H matrices as three-term bf16 split images
 * in streaming order).  Run after every optimiser step, before orl_gt_fwd / orl_gt_bwd. */
int orl_gt_prep(const orl_gt_desc* d, float* image, void* stream);
/* Head outputs of rows i = 0..mb-1 read at x + (idx ? idx[i] : i) * ldx + col0 (D floats each): head_out0 [mb, head_n[0]],
 * head_out1 [mb, head_n[1]] (two heads: the shared network's act / v_out). */
int orl_gt_fwd(const orl_gt_desc* d, const float* image, const float* x, int ldx, int col0, const int64_t* idx, int mb,
               float* head_out0, float* head_out1, void* stream);
/* Backward of the same rows given dhead = d loss / d head outputs [mb, head_n]: the forward is recomputed on chip, the
 * gradient of every parameter the descriptor names is WRITTEN into grad (same offsets as theta).  partials: scratch of
 * >= raw + 24 floats (up to 256 rows of raw + 24 are used: one per workgroup, summed in a fixed order), raw: raw floats. */
int orl_gt_bwd(const orl_gt_desc* d, const float* image, const float* x, int ldx, int col0, const int64_t* idx, int mb,
               const float* dhead0, const float* dhead1, float* partials, int64_t partials_floats, float* raw, float* grad,
               void* stream);
/* The whole update of a minibatch for one tower in ONE launch (+ the reduction of its per-workgroup sums): forward of the
 * rows of the update records, PPOAlgorithm.prepare_loss on the head outputs (openrl/algorithms/ppo.py:178-361 - the same
 * per-row code as orl_gen_policy_loss / orl_gen_value_loss, csrc/orl_gen_loss.h), backward, gradients written into grad;
 * no head output or gradient crosses HBM.  policy_head / value_head: index of the descriptor's head each loss applies to
 * (-1: none); policy_grad = 0 drops the policy head's gradient but keeps its logging sums (turn_on = False on a shared
 * network).  sums_out [24] = {policy-loss sum, entropy sum, ratio sum, -, dlogstd[16], value-loss sum, -, -, -}: what the two
 * loss kernels' partials reduce to.  partials: rows of raw + 24 floats. */
typedef struct orl_gt_loss {
  orl_head_desc head;   /* the policy head's distribution (ignored when policy_head < 0) */
  const float* logstd;  /* Gaussian / mixed heads */
  const float* den;     /* {sum of active masks, rows}: orl_gen_denoms */
  const float* vn_state;/* ValueNorm running state or NULL */
  orl_ppo_hparams hp;
  int32_t Dp, Dc, a_w, K; /* record layout (orl_record_width) */
  int32_t policy_head, value_head, policy_grad, reserved;
} orl_gt_loss;
int orl_gt_train(const orl_gt_desc* d, const float* image, const float* records, int rec_width, int col0,
                 const int64_t* idx, int mb, const orl_gt_loss* loss, float* partials, int64_t partials_floats, float* raw,
                 float* grad, float* sums_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORL_HIP_H */
