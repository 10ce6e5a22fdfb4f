/* imitation_hip.h -- C ABI of libimitation_hip.so (gfx950 / MI355X).
 *
 * The reference (HumanCompatibleAI/imitation) is pure Python and has NO FFI boundary
 * (SURVEY 8b); each entry point below therefore names the reference Python code whose
 * arithmetic it replaces (paths relative to /root/reference/src/imitation/), and
 * INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions: every pointer is a DEVICE pointer to row-major fp32 unless stated; `stream`
 * is a hipStream_t passed as void*; all calls are asynchronous on that stream and return 0
 * on success, a negative IA_ERR_* for bad arguments or a positive hipError_t.
 * No call allocates, frees or synchronises.
 */
#ifndef IMITATION_HIP_H
#define IMITATION_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IA_MAX_LAYERS 8
#define IA_ACT_NONE 0
#define IA_ACT_RELU 1
#define IA_ACT_TANH 2
#define IA_ACT_SOFTPLUS 3

/* Dense stack = util/networks.py:204-283 `build_mlp` without the optional input norm:
 * Linear(dims[0],dims[1]) -act- ... Linear(dims[n-1],dims[n]).  Parameters live in ONE flat
 * buffer in torch `parameters()` order: W0[dims1,dims0], b0[dims1], W1, b1, ...            */
typedef struct {
  int n_layers;
  int dims[IA_MAX_LAYERS + 1];
  int hidden_act;
} ia_mlp_desc;

int ia_version(void);

/* HOST helper (no device work): `count` consecutive `np.random.permutation(n)` draws of NumPy's
 * legacy MT19937 global generator, bit for bit, advancing (key[624], *pos) in place -- the
 * per-epoch minibatch order of [SB3 RolloutBuffer.get] (adversarial/common.py:391-403 -> PPO.train).
 * Lets the host draw them off the Python thread while the rollout runs. out: int64 [count, n]. */
int ia_host_mt19937_permutations(uint32_t* key, int* pos, int64_t n, int count, int64_t* out);
/* ... followed, on a copy of the generator, by `rows` draws of np.random.randint(high, size=row_len) (legacy int64
 * masked rejection): the replay ring's index rows of the round's discriminator updates (`data/buffer.py:366-377`,
 * `algorithms/adversarial/common.py:557-575`). key / pos stay BEHIND THE PERMUTATIONS, key_post / pos_post receive the
 * state behind the rows. */
int ia_host_mt19937_permutations_then_randint(uint32_t* key, int* pos, int64_t n, int count, int64_t* out, int64_t high,
                                              int64_t rows, int64_t row_len, int64_t* out_rows, uint32_t* key_post,
                                              int* pos_post);
/* HOST helper: out[c] = np.random.RandomState(seeds[c, 0:seed_len]).permutation(n) for c < count, each on
 * its own host thread. No reference counterpart (the reference is single-process): the shared-seed
 * minibatch order of the data-parallel PPO update. seeds: uint32 [count, seed_len]. */
int ia_host_mt19937_seeded_permutations(const uint32_t* seeds, int seed_len, int64_t n, int count, int64_t* out);
int64_t ia_mlp_param_count(const ia_mlp_desc* d);
/* floats of workspace per row needed for hidden activations (sum of hidden widths) */
int64_t ia_mlp_hidden_floats_per_row(const ia_mlp_desc* d);

/* Raw fp32 MFMA contraction (test / bench entry). mode 0: C=act(A.B^T+bias); 1: C=(A.B)*act'(P);
 * 2: split-K C_s=A^T.B with optional column sums of A into dbias[splits][M]. */
int ia_gemm_f32(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                int K, const float* bias, int act, const float* P, int ldp, int splits, float* dbias,
                void* stream);
/* mode 0 with the product split along K: slab s of `partials` [splits][M][N] receives split s's products, then
 * C = act(sum of the slabs in split order + bias) -- for layers with few output tiles and a long K (the NatureCNN's
 * 3 136 -> 512 `linear` at rollout / minibatch sizes: [SB3 NatureCNN.linear], `/root/reference/src/imitation/policies/base.py`
 * builds it through SB3's `ActorCriticCnnPolicy`). N and ldc multiples of 4, pointers 16-byte aligned. Deterministic. */
int ia_gemm_f32_nt_splitk(const float* A, int lda, const float* B, int ldb, float* partials, float* C, int ldc, int M, int N,
                          int K, const float* bias, int act, int splits, void* stream);

/* Measurement only (bench.py): when enabled every GEMM launch is bracketed by hipEvents on its
 * launch stream; collect() returns per-kernel totals for the 12 kernels (id = mode*4 + tile
 * config {0:128x128, 1:64x64, 2:128x32, 3:32x128}): elapsed ms, algorithmic flops (2*M*N*K),
 * launch count. At most 8192 launches per window (later launches are not timed). */
int ia_gemm_set_config(int cfg); /* tuning: force a tile config (-1 = automatic selection) */
int ia_prof_enable(int on);
int ia_prof_collect(double* ms, double* flops, long long* launches);

/* rewards/reward_nets.py:441-457 `BasicRewardNet.forward` after concat (+ gail.py:75-83 when
 * out_act=IA_ACT_SOFTPLUS): out[R,dims[n]] = mlp(X). `hidden` receives the post-activation
 * hidden layers ([R,dims1] then [R,dims2] ...; needed by ia_mlp_backward, may be scratch). */
int ia_mlp_forward(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R, float* hidden,
                   float* out, int out_act, void* stream);

/* autograd of the above (what `loss.backward()` does at adversarial/common.py:369):
 * per-split partial parameter gradients `partials[splits][param_count]`; optional dX[R,ldx].
 * `dhidden` is scratch of the same size as `hidden`. */
int ia_mlp_backward(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R,
                    const float* hidden, const float* dOut, float* dhidden, float* partials, int splits,
                    float* dX, void* stream);

/* grads[i] (+)= scale * sum_s partials[s][i]  (gradient accumulation across minibatches,
 * adversarial/common.py:352-369) */
int ia_reduce_partials(const float* partials, int splits, int64_t n, float scale, int accumulate, float* grads,
                       void* stream);
/* ... of a PIECE of every slab: element i of slab k at partials[k * stride + i], i < n <= stride (the generic towers' per-tower
 * gradient stacks reduced straight into the pieces of the flat gradient: [SB3 ActorCriticPolicy] keeps `mlp_extractor.policy_net`,
 * `.value_net`, `action_net`, `value_net` in that order, the kernels want each tower's layers contiguous). */
int ia_reduce_partials_strided(const float* partials, int splits, int64_t n, int64_t stride, float scale, int accumulate,
                               float* grads, void* stream);
/* n_segs <= 16 reductions `ia_reduce_partials(partials[g], splits[g], n[g], 1.0, accumulate = 1, dst[g])` in ONE launch (the
 * NatureCNN policy's backward pass: weight and bias slabs of its six layers -- twelve launches per optimiser step before). */
int ia_reduce_partials_multi(int n_segs, const float* const* partials, const int* splits, const int64_t* n, float* const* dst,
                             void* stream);
/* dst[p][0 .. n[p]) = src[p][0 .. n[p]) for p < n_pieces <= 4 in ONE launch (the towers' contiguous parameter stacks
 * refreshed from the flat parameter vector behind an optimiser step: four device-to-device copies before). */
int ia_copy_pieces(int n_pieces, const float* const* src, float* const* dst, const int64_t* n, void* stream);

/* ia_reduce_partials (accumulate = 0) fused with ia_adam_step: one launch per discriminator update
 * when there is a single minibatch and no cross-rank all-reduce in between. */
int ia_reduce_partials_adam(const float* partials, int splits, int64_t n, float scale, float* grads, float* params,
                            float* exp_avg, float* exp_avg_sq, float beta1, float beta2, float eps,
                            float weight_decay, float step_size, float bc2_sqrt, void* stream);

/* torch.optim.Adam single-tensor step (adversarial/common.py:372; SB3 PPO optimiser):
 * step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) are computed by the caller in double. */
int ia_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                 float beta2, float eps, float weight_decay, float step_size, float bc2_sqrt, void* stream);

/* The same step with its step-dependent scalars kept on the DEVICE, for launch sequences that are captured once and
 * replayed (hipGraph replay of a whole PPO update): `ia_adam_step_scalars` increments the int64 step count *step and
 * writes scalars = {lr / (1 - b1^t), sqrt(1 - b2^t)} (double arithmetic; lr_dev, a device double, replaces `lr` when not
 * NULL so that a learning-rate schedule needs no re-capture); `ia_adam_step_dev` reads them. */
int ia_adam_step_scalars(int64_t* step, double lr, const double* lr_dev, double beta1, double beta2, float* scalars,
                         void* stream);
int ia_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                     float beta2, float eps, float weight_decay, const float* scalars, void* stream);

/* util/networks.py:111-134 `RunningNorm.update_stats` (Chan et al.), count is int32 on device.
 * ws: scratch of at least (2*D*blocks+...) floats, see ia_running_norm_ws_floats. */
int64_t ia_running_norm_ws_floats(int R, int D);
int ia_running_norm_update(const float* X, int ldx, int R, int D, float* mean, float* var, int32_t* count,
                           float* ws, void* stream);
/* Data-parallel form of the update: `partial` writes per-slab (mean, M2) moments of the local
 * batch into ws (ia_running_norm_ws_floats floats); after an all-gather of the ranks' ws buffers,
 * `merge` Chan-merges all `groups` (ranks, `rows_per_group` rows each) and applies the reference
 * update with the global batch -- every rank ends with identical statistics. `ws_ld` is the column
 * count the moments were written with (>= D): a norm over the first D columns of a wider batch can
 * reuse that batch's moments (GAIL: the policy's observation norm reuses the discriminator input's). */
int ia_running_norm_partial(const float* X, int ldx, int R, int D, float* ws, void* stream);
int ia_running_norm_merge(const float* ws_all, int groups, int rows_per_group, int D, int ws_ld, float* mean,
                          float* var, int32_t* count, void* stream);
/* util/networks.py:137-201 `EMANorm.update_stats` from the same slab moments (ia_running_norm_partial; `groups` ranks of
 * `rows_per_group` rows each under data parallelism): inv_learning_rate += decay^num_batches; lr = 1 / inv_learning_rate;
 * mean += lr (b_mean - mean); var += lr (b_var + (1 - lr) (b_mean - mean_old)^2 - var); count += rows; num_batches += 1. */
int ia_ema_norm_merge(const float* ws_all, int groups, int rows_per_group, int D, int ws_ld, float* mean, float* var,
                      int32_t* count, float* inv_learning_rate, int32_t* num_batches, float decay, void* stream);
/* n_seq consecutive ia_running_norm_merge updates (`groups` x `rows_per_group` rows each, moments
 * `seq_stride` floats apart) in order, in one launch: the deferred policy feature-norm updates of a
 * round (adversarial/common.py:606-615 side effect). snapshots (nullable): float
 * [n_seq][2][D], the (mean, var) after each update -- what update k's own `evaluate_actions` normalises
 * with (util/networks.py:79-91: update, then normalise), for AIRL's log pi(a|s) (airl.py:99-119). */
int ia_running_norm_merge_seq(const float* ws_seq, int n_seq, int64_t seq_stride, int groups, int rows_per_group,
                              int D, int ws_ld, float* mean, float* var, int32_t* count, float* snapshots,
                              void* stream);
/* util/networks.py:91: Y = (X-mean)/sqrt(var+eps); columns [D,ldy) of Y are zeroed. */
int ia_running_norm_apply(const float* X, int ldx, int R, int D, const float* mean, const float* var, float eps,
                          float* Y, int ldy, void* stream);

/* adversarial/common.py:592-603 + rewards/reward_nets.py:52-118,441-457: gather rows (int64 idx,
 * NULL = identity) from transition tables and concatenate [state|action|next_state|done] into
 * X[row0+i, :] (fp32, zero padded to ldx). Discrete actions (act_i64 != NULL) are one-hot
 * encoded with width act_dim. dones are bytes (0/1). */
int ia_gather_concat(const float* obs, const float* act_f32, const int64_t* act_i64, const float* next_obs,
                     const uint8_t* dones, const int64_t* idx, int n, int obs_dim, int act_dim, int use_state,
                     int use_action, int use_next_state, int use_done, float* X, int ldx, int row0, void* stream);

/* adversarial/common.py:360-368 + 27-92: BCE-with-logits over R rows whose first n_expert rows
 * are labelled 1 and the rest 0; loss scaled by `scale` (= minibatch/batch); dlogits = dLoss/dlogit.
 * stats[8] = {loss, n_correct, n_correct_expert, n_correct_gen, n_pred_gen, entropy_sum,
 * n_expert, n_gen}. */
int64_t ia_bce_ws_floats(int R);
/* ws: ia_bce_ws_floats(R) floats, zero-initialised once by the caller (the kernel restores the zero). */
int ia_bce_logits(const float* logits, int R, int n_expert, float scale, float* dlogits, float* stats, float* ws,
                  void* stream);

/* ONE call = one discriminator minibatch of adversarial/common.py:352-373 for a BasicRewardNet
 * (GAIL): assemble [rows0 | rows1] from two transition tables, (train-mode) RunningNorm update +
 * normalise, dense-stack forward, BCE-with-logits + statistics, backward, split-K reduction into the
 * flat gradient (accumulate != 0: add to it), and when `adam` != 0 the Adam step fused into that
 * reduction. Every pointer is device memory owned by the caller; nothing is allocated. */
typedef struct {
  const ia_mlp_desc* desc;
  float* params; float* grads; float* exp_avg; float* exp_avg_sq;
  float* norm_mean; float* norm_var; int32_t* norm_count; float norm_eps; int update_norm; /* mean NULL: no norm */
  const float* obs0; const float* act0_f32; const int64_t* act0_i64; const float* next0; const uint8_t* done0;
  const int64_t* idx0; int n0;
  const float* obs1; const float* act1_f32; const int64_t* act1_i64; const float* next1; const uint8_t* done1;
  const int64_t* idx1; int n1;
  int obs_dim, act_dim, use_state, use_action, use_next_state, use_done;
  int n_expert; float loss_scale;
  float* X; float* Xn; int ldx; float* hidden; float* dhidden; float* logits; float* dlogits; float* partials;
  int splits; float* rn_ws; float* bce_ws; float* stats;
  int accumulate; int adam; float beta1, beta2, adam_eps, weight_decay, step_size, bc2_sqrt;
  /* optional: a second RunningNorm over the first pnorm_dim columns updated with the SAME batch
   * moments (the policy feature norm side effect); requires update_norm. */
  float* pnorm_mean; float* pnorm_var; int32_t* pnorm_count; int pnorm_dim;
  /* optional: ia_disc_fused_ws_floats(desc, n0+n1, ldx) floats, ZEROED once by the caller. When set and the
   * stack is D -> H -> H -> 1 with ReLU, H in {128, 256}, D <= 63 (ldx <= 64), the update runs as fused launches
   * (assemble+moments+merge | forward+BCE+head gradient per 64-row tile | input gradient + first-layer weight
   * gradient per tile | second-layer weight gradient | slab reduction + Adam + statistics): the hidden
   * activations of a tile stay in LDS; `hidden` then holds h1 and dh2, `Xn` / `dhidden` are not touched. */
  float* fused_ws;
  /* fused path only: X and rn_ws already hold this update's assembled rows / slab moments
   * (ia_disc_assemble_round), norm_mean / norm_var are the statistics to normalise with (update_norm must be 0),
   * and fused_ws holds current W2T / W1 images (ia_disc_fused_prepare once, then kept by the Adam steps):
   * the update is four launches. */
  int pre_assembled;
  /* fused path, H in {128, 256} only; OPT-IN extension (the reference has no gradient penalty): when gp_e is
   * set (n0 == n1 == n_expert required) the update also carries coef * mean_i (|grad_x D(x_hat_i)|_2 - target)^2 with
   * x_hat_i = e_i x_expert_i + (1 - e_i) x_gen_i (rows of X, normalised with the statistics the forward used), in three
   * more tile launches + one split-K product whose slabs join the same reduction (+ Adam). gp_ws:
   * ia_disc_fused_gp_ws_floats(desc, n0, ldx) floats ZEROED once by the caller; gp_out[0] = the mean penalty. */
  const float* gp_e; float gp_coef; float gp_target; float* gp_ws; float* gp_out;
} ia_disc_step_args;
int ia_disc_step_basic(const ia_disc_step_args* a, void* stream);
/* The n updates of one round (`for _ in range(n_disc_updates_per_round): train_disc()`, adversarial/common.py:454-458)
 * in one host call: update k = ia_disc_step_basic(&a[k], stream), in order; stops at the first error. */
int ia_disc_round_basic(const ia_disc_step_args* a, int n, void* stream);
/* 0 when the fused path does not cover the shape (the call then runs the general path). */
int64_t ia_disc_fused_ws_floats(const ia_mlp_desc* d, int R, int ldx);
/* 0 when the fused gradient penalty does not cover the shape (B = interpolated rows = expert rows of an update). */
int64_t ia_disc_fused_gp_ws_floats(const ia_mlp_desc* d, int B, int ldx);
/* A whole round's batch assembly in ONE launch (fused shapes): for k < n_updates, rows idx0 + k*idx_stride /
 * idx1 + k*idx_stride of a's two tables -> X + k*x_stride, RunningNorm slab moments -> rn_ws + k*rn_stride
 * (strides in elements; no statistics are touched: apply them with ia_running_norm_merge_seq, whose snapshots
 * are what update k normalises with -- util/networks.py:79-91 update-then-normalise, one update at a time). */
int ia_disc_assemble_round(const ia_disc_step_args* a, int n_updates, int64_t idx_stride, int64_t x_stride,
                           int64_t rn_stride, void* stream);
/* W2T / padded W1 images of the CURRENT parameters into fused_ws (before a batch of pre_assembled updates). */
int ia_disc_fused_prepare(const ia_mlp_desc* d, const float* params, int R, int ldx, float* fused_ws, void* stream);
/* Measurement only: when set to a device buffer of 16 int64, block 0 of the fused forward / backward tile
 * kernels stores the shader clock at its phase boundaries in [0..7] / [8..12] (NULL switches it off). */
int ia_disc_fused_debug_timing(void* device_buffer_16xi64);
/* Tuning: rows per workgroup of the fused tile kernels, 64 (default: one 512-thread workgroup per CU) or 32
 * (two 256-thread workgroups per CU). */
int ia_disc_fused_tile_rows(int rows);
/* Tuning / measurement: 1 = the update's forward and backward tile passes as two launches (disc_fwd_kernel,
 * disc_bwd_kernel); 0 (default) = both in one (disc_fb_kernel). Outputs are bit-identical. */
int ia_disc_fused_split_tiles(int on);
/* Tuning / measurement: 1 (default) = the part of the update's closing slab reduction + Adam step that does not depend
 * on the split-K product dW2 (first / last layer, b2, the statistics row) runs as the leading workgroups of that
 * product's launch; 0 = the whole reduction in its own launch behind the product. Outputs are bit-identical. */
int ia_disc_fused_side_reduce(int on);
/* Tuning / measurement: form of the one-launch gradient-penalty pass on the 256-wide stack with rows of up to 24 floats: 8
 * (default: one 32-row tile per 512-thread workgroup, its columns over eight waves -- two waves per SIMD, one workgroup per tile),
 * 2 (two tiles per 512-thread workgroup sharing the weight stream: half as many workgroups) or 1 (one tile, four waves: one
 * wave per SIMD, the form before round 5). Same values. With 8 the penalty's pass shares ONE launch with the update's own
 * tile pass (64-row tiles: two independent one-tile 512-thread workgroup bodies, `disc_fb_gp_kernel`); 80 = the eight-wave
 * form as a launch of its own. */
int ia_disc_fused_gp_groups(int groups);
/* Prediction on the fused tile kernel: out[r] = out_act(MLP(normalise(X[r, :D]))) for R assembled rows of a D -> H -> H -> 1
 * ReLU stack (D <= 24, H = 128 / 256; or the reference's default H = 32 with D <= 64: the row kernel, one launch) -- `RewardNet.predict_th` of a whole rollout tile (rewards/reward_nets.py:176-204),
 * i.e. the reward relabelling behind a rollout's last step (rewards/reward_wrapper.py:110-115; GAIL: out_act = IA_ACT_SOFTPLUS,
 * algorithms/adversarial/gail.py:75-83) -- in two launches, the hidden activations never leaving LDS; replaces
 * ia_running_norm_apply + ia_mlp_forward on these shapes (the same fp32 MFMA products; the values differ from that path's by
 * <= 1e-7 relative, tests/test_disc_fused_gpu.py). norm_mean == NULL: no input
 * normalisation. predict_ws: ia_disc_fused_predict_ws_floats(d, ldx) floats (0 = shape not covered; IA_ERR_UNSUPPORTED). */
int64_t ia_disc_fused_predict_ws_floats(const ia_mlp_desc* d, int ldx);
int ia_disc_fused_predict(const ia_mlp_desc* d, const float* params, const float* X, int ldx, int R, const float* norm_mean,
                          const float* norm_var, float norm_eps, int out_act, float* predict_ws, float* out, void* stream);

/* Gradient penalty on the discriminator (OPT-IN extension, default off: BASELINE.json config 3 / the north star name
 * it, the reference has none -- SURVEY M1): E[(|grad_x D(x_hat)|_2 - target)^2] at x_hat = e x_expert + (1-e) x_gen.
 * `ia_gp_interpolate`: rows r < B of X ([expert | generator], 2B rows) -> normalised interpolates Xn[B, ld]
 * (statistics frozen; mean NULL: none). `ia_gp_row_coeffs`: from gn = dD/dXn (ia_mlp_backward with dOut = 1) the
 * per-row penalty pen[B] and Cn = d(coef/B * sum pen)/d gn [B, ld]; the parameter gradient follows from Cn with
 * ia_gemm_f32 calls (ReLU stacks: the masks are locally constant), see imitation_amd/grad_penalty.py. */
int ia_gp_interpolate(const float* X, int ldx, int B, int D, const float* e, const float* mean, const float* var,
                      float eps, float* Xn, int ld, void* stream);
int ia_gp_row_coeffs(const float* gn, int ld, int B, int D, const float* var, float eps, float coef, float target,
                     float* Cn, float* pen, void* stream);
/* The same for AIRL's shaped reward f(s, a, s') = g([s | a | s' | d]) + gamma (1 - d) h(s') - h(s)
 * (rewards/reward_nets.py:727-733; d = the interpolated done flag dhat[B], constant): from the three stacks' input
 * gradients gn_b[B, ldb] (base, blocks as flagged), gn_n / gn_c[B, ldp] (potential at s' / at s) the penalty of
 * |grad_(s, a, s', d) f|_2 per row and the coefficients Cn_b / Cn_n / Cn_c each stack's second pass starts from. */
int ia_gp_shaped_coeffs(const float* gn_b, int ldb, const float* gn_n, const float* gn_c, int ldp, const float* dhat, int B,
                        int obs_dim, int act_dim, int use_state, int use_action, int use_next_state, int use_done,
                        const float* var_b, float eps_b, const float* var_p, float eps_p, float gamma, float coef,
                        float target, float* Cn_b, float* Cn_n, float* Cn_c, float* pen, void* stream);

/* adversarial/airl.py:118 + rewards/reward_nets.py:701-736:
 * logits = g + gamma*(1-done)*h_next - h_cur - logp ; and the matching dOut routing. */
int ia_airl_logits(const float* g, const float* h_cur, const float* h_next, const float* dones /*0/1 fp32*/,
                   const float* logp, float gamma, int R, float* logits, void* stream);
int ia_airl_route_grad(const float* dlogits, const float* dones, float gamma, int R, float* dg, float* dh_cur,
                       float* dh_next, void* stream);

/* ONE AIRL discriminator update for the scripts' default shaped reward net (adversarial/airl.py:99-132,
 * rewards/reward_nets.py:674-809: reward MLP Db -> 32 -> 1 on [s | a | s' | done], potential MLP Dp -> 32 -> 32 -> 1 on
 * s' and s, ReLU) without the ~40 launches of the stack-by-stack path (csrc/airl_fused.hip): one MFMA kernel over the
 * rows (normalise with the given statistics -- potential: `A` after the next-state update, `B` after the state update
 * -- three forwards, logits = g + gamma (1-done) h(s') - h(s) - log pi, BCE + statistics, deltas) and the three
 * hidden-layer weight-gradient GEMMs. `partials` = [ia_airl_fused_slabs(R)][n_params] split-K slabs (base stack's
 * parameters first, torch order) for ia_reduce_partials(_adam). Returns -2 for other geometries (ia_airl_fused_ok:
 * widths 32, input widths <= 64). */
typedef struct ia_adam_args {   /* torch.optim.Adam step over one flat buffer, see ia_adam_step */
  float* grads;                 /* [n] receives the reduced gradient */
  float* exp_avg;
  float* exp_avg_sq;
  float beta1, beta2, eps, weight_decay, step_size /* lr / (1 - b1^t) */, bc2_sqrt /* sqrt(1 - b2^t) */;
} ia_adam_args;
/* Data-parallel tail of a fused BasicRewardNet update (adversarial/common.py:352-373 on the union of the
 * ranks' batches): ia_disc_step_basic with adam = 0 leaves this rank's reduced gradient in adam->grads, the caller sums it
 * over the ranks with ONE all-reduce, then this launch scales it by grad_scale (1 / world) in place, applies
 * torch.optim.Adam's step to `params` and refreshes the W2T / W1 images of fused_ws for the next pre_assembled update. */
int ia_disc_fused_adam(const ia_mlp_desc* d, float* params, float grad_scale, int R, int ldx, float* fused_ws,
                       const ia_adam_args* adam, void* stream);
int ia_airl_fused_ok(int Db, int Dp, int hb, int hp1, int hp2);
int ia_airl_fused_slabs(int R);
/* measurement: shader-clock stamps of the row kernel's phases (workgroup 0) into buf (>= 16 int64; NULL: off) */
int ia_airl_debug_timing(long long* buf);
int ia_airl_step_shaped(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp,
                        const float* dones, const float* logp, const float* bmean, const float* bvar, float beps,
                        const float* pmeanA, const float* pvarA, const float* pmeanB, const float* pvarB, float peps,
                        const float* params_base, const float* params_pot, float gamma, float scale, int R, int n_expert,
                        float* Ab, int ldab, float* Db1, float* Ap, int ldap, float* H1, float* Dp1, float* Dp2,
                        float* partials, float* logits, float* stats, float* bce_part, unsigned* ticket,
                        const ia_adam_args* adam /* nullable: reduce the slabs + Adam step in the same call */,
                        void* stream);

/* Batch assembly of one AIRL update in ONE launch (adversarial/common.py:564-603, rewards/reward_nets.py:441-457): rows
 * idx0 (n0 of them, expert; null idx = the first n) then idx1 (n1, generator) of two transition tables -> Xb[n0+n1, ldb]
 * ([state | action, one-hot when act_i64 | next state | done] as flagged), Sn / Sc[., ldp] (next observations,
 * observations) and dones[.] as 0/1 floats; with ws_* given, also the RunningNorm slab moments of each matrix
 * ([ceil(R/256)][2][D]: layout and arithmetic of ia_running_norm_partial); with pol_obs / pol_act given, the unpadded
 * observation / action rows the generator policy's log pi(a|s) is evaluated on (common.py:606-615). Padding columns
 * are not written. */
int ia_airl_prepare(const float* obs0, const float* act0_f32, const int64_t* act0_i64, const float* next0,
                    const uint8_t* done0, const int64_t* idx0, int n0, const float* obs1, const float* act1_f32,
                    const int64_t* act1_i64, const float* next1, const uint8_t* done1, const int64_t* idx1, int n1,
                    int obs_dim, int act_dim, int use_state, int use_action, int use_next_state, int use_done, float* Xb,
                    int ldb, float* Sn, float* Sc, int ldp, float* dones, float* ws_b, float* ws_n, float* ws_c,
                    float* pol_obs /*[R, obs_dim], nullable*/, float* pol_act /*[R, act_dim] or [R] index, nullable*/,
                    void* stream);
/* The train-mode RunningNorm updates of one shaped-net forward (util/networks.py:111-134 in reward_nets.py:708-710's
 * order) from ia_airl_prepare's slab moments, one launch: base input norm (ws_b null: skipped); potential input norm
 * (ws_n, ws_c null: skipped) with the next-state batch -- (mean, var) then copied to snapA[2][Dp] -- and with the state
 * batch. Data parallelism: `groups` ranks contribute R rows each, their slab moments `group_stride` floats
 * apart inside the all-gathered buffer (0: back to back); all are merged in rank order, i.e. the update of ONE process on
 * the concatenated batch. `ticket`: one zeroed word (left zeroed). */
int ia_airl_stats_merge(const float* ws_b, const float* ws_n, const float* ws_c, int groups, int64_t group_stride, int R,
                        int Db, int Dp, float* bmean, float* bvar, int32_t* bcount, float* pmean, float* pvar,
                        int32_t* pcount, float* snapA, unsigned* ticket, void* stream);

/* Gradient penalty of AIRL's shaped reward (OPT-IN extension, see ia_gp_shaped_coeffs) for the geometry of
 * ia_airl_fused_ok, on the batches ia_airl_prepare assembled (Xb, Sn, Sc: [2B, ld] = [expert | generator] rows, dones[2B];
 * e[B] interpolation weights; statistics frozen, null: none): coef * mean_i (|grad_(s,a,s',d) f|_2 - target)^2, its
 * parameter gradient ADDED to grads ([base | potential] flat layout). One MFMA row kernel (forward masks, input
 * gradients, row coefficients, second pass), three split-K weight-gradient GEMMs, one accumulate. Workspaces: U1b[B,32],
 * Cb[B,ldb], U1p / U2p / V1p[2B,32], Cp[2B,ldp], partials[ia_airl_fused_slabs(B)][n_params] (ZEROED once by the caller),
 * pen_part[slabs], ticket (one zeroed word). pen_out[0] = mean (|grad f| - target)^2. */
int ia_airl_gp_shaped(const float* Xb, int ldb, int Db, const float* Sn, const float* Sc, int ldp, int Dp,
                      const float* dones, const float* e, const float* bmean, const float* bvar, float beps,
                      const float* pmean, const float* pvar, float peps, const float* params_base,
                      const float* params_pot, int obs_dim, int act_dim, int use_state, int use_action,
                      int use_next_state, int use_done, float gamma, float coef, float target, int B, float* U1b,
                      float* Cb, float* U1p, float* Cp, float* U2p, float* V1p, float* partials, float* pen_part,
                      float* pen_out, unsigned* ticket, float* grads, void* stream);

/* rewards/reward_nets.py:637-671 `NormalizedRewardNet.predict_processed` applied once per env
 * step: out[t,:] = (raw[t,:]-mean)/sqrt(var+eps) with the statistics of steps < t, then (when
 * update_stats) the Chan update with raw[t,:]. mean/var are 1-element buffers, count int32. */
int ia_reward_norm_sequential(const float* raw, int T, int n, float eps, int update_stats, float* mean, float* var,
                              int32_t* count, float* out, void* stream);
/* Data-parallel form: every rank relabels its own env batch [T, n]; `ia_reward_step_moments` leaves the
 * per-step (mean, M2) of its raw rewards in moments[T][2]; after an all-gather (rank-major [groups][T][2]),
 * `ia_reward_norm_sequential_groups` walks the steps like ia_reward_norm_sequential but absorbs, per step, the batch of
 * ALL ranks (groups * n samples) -- the statistics of one process on the env batches side by side. */
int ia_reward_step_moments(const float* raw, int T, int n, float* moments, void* stream);
int ia_reward_norm_sequential_groups(const float* raw, int T, int n, float eps, int update_stats, const float* moments_all,
                                     int groups, float* mean, float* var, int32_t* count, float* out, void* stream);

/* generic row gather: dst[i,:] = src[idx[i],:] (width floats) */
int ia_gather_rows(const float* src, const int64_t* idx, int n, int width, float* dst, void* stream);

/* ---- generator (policy / PPO) kernels: see policy section below (policy.hip) ---- */

/* SB3 ActorCriticPolicy with net_arch=[H,H] tanh towers (policies/base.py:92-104), flat
 * parameter order = torch parameters(): [log_std (Box only)], pi.W1,b1,W2,b2, vf.W1,b1,W2,b2,
 * action_net.W,b, value_net.W,b. */
typedef struct {
  int obs_dim;
  int act_dim;      /* Box: action dims; Discrete: number of actions */
  int hidden;       /* 32 or 64 */
  int discrete;     /* 0 Box / DiagGaussian, 1 Discrete / Categorical */
  int has_norm;     /* NormalizeFeaturesExtractor(RunningNorm) in front (policies/base.py:123-149) */
  float norm_eps;
} ia_policy_desc;

int64_t ia_policy_param_count(const ia_policy_desc* d);
/* params_t = same flat layout with the four tower matrices stored [in][out] (kept in sync by
 * ia_ppo_minibatch; call this after loading parameters from the host). */
int ia_policy_transpose(const ia_policy_desc* d, const float* params, float* params_t, void* stream);

/* [SB3 ActorCriticPolicy.forward] rollout step (collect_rollouts):
 * Box: actions = mean + exp(log_std)*noise, clipped = clip(actions, low, high);
 * Discrete: `noise` holds one uniform(0,1) per row, inverse-CDF sampling over softmax
 * (actions / clipped are then fp32 action indices [n]); a negative entry selects the mode
 * (argmax, first index on ties) for `predict(deterministic=True)`. */
int ia_policy_act(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                  const float* norm_var, const float* obs, int n, const float* noise, const float* low,
                  const float* high, float* actions, float* clipped, float* values, float* logp, void* stream);
/* [SB3 `collect_rollouts`'s per-step `policy.forward`] for a whole rollout in ONE resident launch: the workgroups take
 * step t when `ready[0]` (int32 in pinned, device-mapped host memory) reaches t + 1 -- the host posts it after the
 * step's observations / noise are in their pinned tiles --, run the body of `ia_policy_act` on the step's tiles (strides
 * in floats between consecutive steps; 0 = the same tile every step) and acknowledge in `done[workgroup]` (pinned host
 * memory, ceil(n / 64) ints) once the step's outputs, the clipped actions in host memory first of all, have left.
 * Bounded: `ready[0] < 0` (abort) or `timeout_s` without a new step end the kernel; `done` then holds -(t + 1).
 * last_val (may be NULL): when set, the host may post one more step, T (`ready[0] = T + 1` once `obs + T * s_obs`
 * holds the observation behind the last step): the kernel then writes V of those rows there (the GAE bootstrap of
 * [SB3 collect_rollouts]) and acknowledges with T + 1; nobody has to wait for that on the host.
 * IA_ERR_UNSUPPORTED (-> launch `ia_policy_act` per step) for shapes the matrix-core act kernel does not cover. */
int ia_policy_rollout_mailbox(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                              const float* norm_var, int n, const float* low, const float* high, const float* obs,
                              int64_t s_obs, const float* noise, int64_t s_noise, float* actions, int64_t s_act,
                              float* clipped, int64_t s_clip, float* values, int64_t s_val, float* logp, int64_t s_lp,
                              float* last_val, int T, const int32_t* ready, int32_t* done, double timeout_s,
                              void* stream);
/* The same for the host-sampled Discrete step (`ia_policy_logits` per step + torch.multinomial on the host): step t's
 * logits land in the SAME pinned [n, A] tile every step (the host samples from it before posting the next step), the
 * values in `values + t * s_val`. */
int ia_policy_logits_mailbox(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                             const float* norm_var, int n, const float* obs, int64_t s_obs, float* logits, float* values,
                             int64_t s_val, int T, const int32_t* ready, int32_t* done, double timeout_s, void* stream);
/* Host half: spin (no GIL under ctypes) until all `n` flags have reached `target`. 0 reached, 1 timed out, -1 a flag is
 * negative (the kernel gave up). */
int ia_host_wait_i32(const volatile int32_t* flags, int n, int target, double timeout_s);

/* [SB3 evaluate_actions / predict_values] without grad (adversarial/common.py:490-496): any of
 * logp/values/entropy may be NULL. actions: fp32 [n,act_dim] (Box) or fp32 action index [n]. */
int ia_policy_evaluate(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                       const float* norm_var, const float* obs, const float* actions, int n, float* logp,
                       float* values, float* entropy, void* stream);

/* Slab moments of the observation columns of a round's n_updates batches in one launch (the policy's train-mode feature
 * RunningNorm side effect of train_disc's log pi(a|s) evaluation, common.py:606-615, taken ahead of the updates): batch k =
 * rows idx0 + k*idx_stride (n0 of the first table) then idx1 + k*idx_stride (n1 of the second); X [n_updates][n0+n1][ldx]
 * scratch; rn_ws + k*rn_stride <- batch k's moments (ia_running_norm_partial's layout and arithmetic; feed
 * ia_running_norm_merge_seq). */
int ia_obs_moments_round(const float* obs0, const int64_t* idx0, int n0, const float* obs1, const int64_t* idx1, int n1,
                         int obs_dim, int n_updates, int64_t idx_stride, float* X, int ldx, float* rn_ws, int64_t rn_stride,
                         void* stream);

/* One AIRL discriminator update on the fused shaped-net path as ONE record -- the arguments of the four calls it is made
 * of, in their order: ia_airl_prepare (batch assembly incl. the policy's rows), ia_airl_stats_merge (train-mode input
 * statistics; skipped when ws_b, ws_n and ws_c are all NULL), ia_policy_evaluate (log pi(a|s) of the assembled rows under the
 * statistics pol_norm_mean / pol_norm_var: common.py:606-615), ia_airl_step_shaped (forward, BCE, backward, reduction,
 * Adam) -- with the opt-in gradient penalty: reduction, ia_airl_gp_shaped and the Adam step as three more calls. Field
 * names are the parameters' names of those entries. */
typedef struct {
  /* ia_airl_prepare */
  const float* obs0; const float* act0_f32; const int64_t* act0_i64; const float* next0; const uint8_t* done0;
  const int64_t* idx0; int n0;
  const float* obs1; const float* act1_f32; const int64_t* act1_i64; const float* next1; const uint8_t* done1;
  const int64_t* idx1; int n1;
  int obs_dim, act_dim, use_state, use_action, use_next_state, use_done;
  float* Xb; int ldb; float* Sn; float* Sc; int ldp; float* dones;
  float* ws_b; float* ws_n; float* ws_c; float* pol_obs; float* pol_act;
  /* ia_airl_stats_merge (one rank: groups = 1) */
  int Db, Dp;
  float* bmean; float* bvar; int32_t* bcount; float* pmean; float* pvar; int32_t* pcount; float* snapA;
  unsigned* merge_ticket;
  /* ia_policy_evaluate */
  const ia_policy_desc* pol; const float* pol_params; const float* pol_params_t; const float* pol_norm_mean;
  const float* pol_norm_var; float* logp;
  /* ia_airl_step_shaped (Xb ... dones, logp as above) */
  const float* f_bmean; const float* f_bvar; float beps;
  const float* pmeanA; const float* pvarA; const float* pmeanB; const float* pvarB; float peps;
  const float* params_base; const float* params_pot; float gamma; float scale; int n_expert;
  float* Ab; int ldab; float* Db1; float* Ap; int ldap; float* H1; float* Dp1; float* Dp2;
  float* partials; float* logits; float* stats; float* bce_part; unsigned* ticket;
  ia_adam_args adam;   /* the reduction + Adam step of this update (its step_size / bc2_sqrt) */
  /* opt-in gradient penalty (gp_e != NULL; n0 == n1 required): ia_airl_step_shaped then leaves the slabs, they are
   * reduced into adam.grads (ia_reduce_partials), ia_airl_gp_shaped ADDS the penalty's gradient (its parameters' names
   * below), and ia_adam_step over the n_params parameters at params_base closes the update */
  const float* gp_e; float gp_coef; float gp_target; int n_slabs; int64_t n_params;
  float* U1b; float* Cb; float* U1p; float* Cp; float* U2p; float* V1p; float* gp_partials; float* pen_part; float* pen_out;
  unsigned* gp_ticket;
} ia_airl_update_args;
/* The n updates of one round (`for _ in range(n_disc_updates_per_round): train_disc()`, adversarial/common.py:454-458)
 * for AIRL's fused shaped-net update, in one host call: update k = the four calls above with a[k], in order; stops at the
 * first error. */
int ia_airl_round(const ia_airl_update_args* a, int n, void* stream);


/* Head outputs only: logits[n, act_dim] = action_net(latent_pi) (Categorical logits / Gaussian means) and,
 * when values != NULL, the value head. For host-side sampling with the reference's own RNG call:
 * [SB3 CategoricalDistribution.sample] = torch.multinomial on torch's global CPU generator
 * (adversarial/common.py:414-419 -> collect_rollouts). */
int ia_policy_logits(const ia_policy_desc* d, const float* params, const float* params_t, const float* norm_mean,
                     const float* norm_var, const float* obs, int n, float* logits, float* values, void* stream);

/* [SB3 RolloutBuffer.compute_returns_and_advantage]: arrays are [T,n] fp32. */
int ia_gae(const float* rewards, const float* values, const float* episode_starts, const float* last_values,
           const float* last_dones, int T, int n, float gamma, float gae_lambda, float* advantages,
           float* returns, void* stream);

/* The tail of a rollout whose reward is a fused-shape discriminator (GAIL: softplus of the logit) in ONE host call: the
 * launches of ia_gather_concat (identity rows of the [T, n] tile) + ia_disc_fused_predict + the rewards' copy to a pinned
 * host tile (rewards/reward_wrapper.py:117-133 reads them there; may be NULL) + ia_gae, in that order -- what
 * [SB3 collect_rollouts] does behind its last step (rewards/reward_wrapper.py:110-115 once per step == once on the tile)
 * followed by [SB3 RolloutBuffer.compute_returns_and_advantage]. Replaces four host calls and the Python between them. */
typedef struct {
  const float* obs; const float* act_f32; const int64_t* act_i64; const float* next_obs; const uint8_t* dones;
  int obs_dim, act_dim, use_state, use_action, use_next_state, use_done;
  float* X; int ldx;                                   /* assembled rows [T * n, ldx] */
  const ia_mlp_desc* desc; const float* params; const float* norm_mean; const float* norm_var; float norm_eps; int out_act;
  float* predict_ws;                                   /* ia_disc_fused_predict_ws_floats(desc, ldx) floats */
  float* rewards; float* rewards_host;                 /* [T, n] device tile; pinned host copy or NULL */
  const float* values; const float* episode_starts; const float* last_values; const float* last_dones;
  int T, n; float gamma, gae_lambda; float* advantages; float* returns;
} ia_rollout_tail_args;
int ia_rollout_tail(const ia_rollout_tail_args* a, void* stream);

/* [SB3 collect_rollouts] rewards[i] += gamma * V(terminal_obs_i) where truncated[i]. */
int ia_timeout_bootstrap(float* rewards, const float* terminal_values, const uint8_t* truncated, float gamma,
                         int64_t n, void* stream);

/* One PPO minibatch (SB3 PPO.train body): rows `idx` (int64 into the env-major
 * flattened rollout, row = env*T+t) of time-major [T,n_envs,...] obs/actions/old_logp/advantages/
 * returns. Three launches: (1) minibatch advantage mean/std (unbiased) + RunningNorm update of the
 * policy's feature norm when in train mode; (2) forward, clipped-surrogate/value/entropy loss,
 * backward, per-wave partial gradients; (3) fixed-order reduction, clip_grad_norm_, Adam, refresh
 * of params_t. stats[8] = {pg_loss, value_loss, entropy_loss, approx_kl, clip_fraction, loss,
 * grad_norm, clip_coef}. */
/* gather_rows = rows gathered per call: `batch` for the single-minibatch entries, T*n_envs for
 * ia_ppo_epoch (which copies the whole permuted epoch into contiguous rows first). */
int64_t ia_ppo_ws_floats(const ia_policy_desc* d, int batch, int64_t gather_rows);
int ia_ppo_minibatch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                     int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                     const float* old_logp, const float* advantages, const float* returns, const int64_t* idx,
                     int batch, int T, int n_envs, int normalize_adv, float clip_range, float ent_coef,
                     float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, float beta1,
                     float beta2, float adam_eps, float step_size, float bc2_sqrt, float* ws, float* stats,
                     void* stream);
/* Data-parallel split (one rank per GPU): `_grad` leaves the rank's minibatch gradient (mean-loss
 * gradient of its local rows) at ws + ia_ppo_grad_offset() floats; the caller all-reduces (mean)
 * that flat bucket over RCCL; `_apply` then clips by the global norm and steps Adam. */
int ia_ppo_minibatch_grad(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                          int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                          const float* old_logp, const float* advantages, const float* returns, const int64_t* idx,
                          int batch, int T, int n_envs, int normalize_adv, float clip_range, float ent_coef,
                          float vf_coef, float* ws, void* stream);
int64_t ia_ppo_grad_offset(const ia_policy_desc* d, int batch);
/* Debug/measurement: when set to a device buffer of 32 int64, block 0 of every ppo_grad launch
 * stores the shader clock at its phase boundaries in [0..11]; ia_ppo_update accumulates 100 MHz ticks
 * per step phase in [0..7] and stores the last step's phase clocks in [16..27] (NULL switches it off). */
int ia_ppo_debug_timing(void* device_buffer_16xi64);
/* Tuning / measurement: 1 = `ia_ppo_epoch` keeps two launches per minibatch for 64-wide towers (default 0: one launch
 * per epoch, the minibatch steps as phases of a co-resident grid; a row block's two towers
 * are two four-wave workgroups with the tower's parameters resident in LDS when that fits); 2 = one launch per epoch
 * with whole row-block workgroups (eight waves, both towers, weight fragments from memory: the form before);
 * 3 = the one-tower kernel with grid barriers between the phases (default 0 hands the slabs, the partial sums of
 * squares and the new parameters over as 8-byte value / sequence words instead: no barrier; bit-identical results);
 * 4 = that word-exchange kernel with round 4's gradient body (`ppo_epoch_ll_kernel`) also where the default applies:
 * observation widths <= 32 run `ppo_epoch_ll2_kernel` -- the same word exchange (chunk owners sum, clip and step their
 * chunk of the parameters and hand the new words on), with the gradient phase on the transposed register-resident layer
 * chain of the 32-wide persistent kernel (activations in registers from x to dz1, weight fragments as one ds_read_b128
 * of a padded LDS image; [SB3 PPO.train] on the default MlpPolicy). Wider observations keep `ppo_epoch_ll_kernel`;
 * minibatches of more than 32 row blocks (or towers whose images do not fit in LDS) the barrier form (3).
 * Round 6, inside `ppo_epoch_ll2_kernel`: minibatches of up to 1 024 rows run on 32-ROW blocks (eight waves: two row groups x
 * four quarters of every layer's output features; twice the workgroups, half the chain and half the weight-gradient tiles per
 * compute unit), larger ones on 64-row blocks with eight waves per tower workgroup (feature halves, two waves per SIMD);
 * 6 = the eight-wave 64-row form also where the 32-row blocks apply; 7 = 32-row blocks on four waves (feature halves: the same
 * MFMAs in the same order per tile as the default's quarters -- bit-identical gradients). (Round 5's four-wave 64-row form, to
 * which 6 was bit-identical in that sense, was retired in round 6.) */
int ia_ppo_epoch_split(int on);
/* Measurement only: device buffer of 64 int64 (NULL: off); workgroup 0 of the one-launch-per-epoch kernel accumulates
 * 100 MHz ticks per phase in [0..5] = {gradient, barrier, slab sum, barrier, norm + Adam, barrier}; [16..27] / [32..43]:
 * shader-clock stamps inside the last step's gradient phase (row block 0; policy / value tower workgroup). */
int ia_ppo_epoch_debug_timing(void* device_buffer_64xi64);
int ia_ppo_minibatch_apply(const ia_policy_desc* d, float* params, float* params_t, int batch, float ent_coef,
                           float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, float beta1,
                           float beta2, float adam_eps, float step_size, float bc2_sqrt, float* ws, float* stats,
                           void* stream);
/* ---- image policies (SB3 NatureCNN / ActorCriticCnnPolicy; algorithms/bc.py:94-156 with BASELINE config 4) ----
 * Convolution = im2col + ia_gemm_f32 (mode 0, weights [Cout, Cin*KH*KW] as torch stores them); activations
 * are channel-last [B, H, W, C]; row m = (b*OH + oh)*OW + ow. `_u8_nchw`: the policy's input frames, uint8
 * channel-first ([SB3 preprocess_obs]: x / 255 -> scale), column index k = (c*KH + i)*KW + j (torch's weight
 * layout as stored). `_f32_nhwc` / `ia_col2im_nhwc`: column index k = (i*KW + j)*C + c, i.e. weights kept as
 * [Cout, KH, KW, Cin] (both sides of the copy are then contiguous runs). */
int ia_im2col_u8_nchw(const uint8_t* x, int B, int C, int H, int W, int KH, int KW, int S, float scale, float* col,
                      void* stream);
int ia_im2col_f32_nhwc(const float* x, int B, int H, int W, int C, int KH, int KW, int S, float* col, void* stream);
/* Input gradient of a convolution from its column gradient (gather form, fixed order, no atomics);
 * relu_mask (nullable) = that input's own post-ReLU value: the gradient is zeroed where it is <= 0. */
int ia_col2im_nhwc(const float* dcol, int B, int H, int W, int C, int KH, int KW, int S, const float* relu_mask,
                   float* dx, void* stream);
/* Padded convolutions of the reward CNN (util/networks.py:286-357 `build_cnn`, rewards/reward_nets.py:460-610
 * `CnnRewardNet`): im2col / col2im with a zero border of P pixels resolved in the index arithmetic (column index
 * k = (i*KW + j)*C + c, activations channel-last); OH = (H + 2P - KH)/S + 1. */
int ia_im2col_f32_nhwc_pad(const float* x, int B, int H, int W, int C, int KH, int KW, int S, int P, float* col,
                           void* stream);
int ia_col2im_nhwc_pad(const float* dcol, int B, int H, int W, int C, int KH, int KW, int S, int P, float* dx,
                       void* stream);
/* out = y > 0 ? dy : 0 (ReLU backward from the post-activation value). */
int ia_relu_backward(const float* dy, const float* y, int64_t n, float* out, void* stream);
/* nn.AdaptiveAvgPool2d(1) on channel-last activations y[B, HW, C] -> out[B, C], and its backward (dy = dout / HW). */
int ia_avgpool_nhwc(const float* y, int B, int HW, int C, float* out, void* stream);
int ia_avgpool_nhwc_backward(const float* dout, int B, int HW, int C, float* dy, void* stream);
/* The reward CNN's geometry (3 x 3, stride 1, "same" padding, 32 -> 32 channels; `rewards/reward_nets.py:460-610` with
 * `util/networks.py:286-357`'s defaults), channel-last tensors dz[B, H, W, 32], x[B, H, W, 32]: weight and bias gradient as
 * ia_conv3x3_c32_wgrad_slabs(B) slabs part[slabs][32][3][3][32] (torch's [Cout, KH, KW, Cin] per slab), dbp[slabs][32] -- reduce
 * with ia_reduce_partials. A workgroup owns whole images, the 32 x 288 gradient stays in its accumulators, x and dz are read
 * once. -2 (IA_ERR_UNSUPPORTED) for W > 128. */
int ia_conv3x3_c32_wgrad_slabs(int B);
int ia_conv3x3_c32_wgrad(const float* dz, const float* x, int B, int H, int W, float* part, float* dbp, void* stream);
/* Forward of that 32 -> 32 layer, y[B, H, W, 32] = act(bias + conv3x3(x, w)) with the weights handed over TRANSPOSED,
 * wt[ky][kx][ci][co] (`bias` nullable; relu != 0: ReLU; `mask` nullable, laid out like y: outputs zeroed where mask <= 0) -- and,
 * called with wt[ky][kx][co][ci] = W[co][2 - ky][2 - kx][ci] on dz, its input gradient (mask: the ReLU output of the layer below). A workgroup
 * takes 8 output rows of one image with their 10 input rows and the weights in LDS. -2 when a row band does not fit 160 KB of LDS
 * (W > 94). */
int ia_conv3x3_c32_conv(const float* x, const float* wt, const float* bias, const float* mask, int B, int H, int W, int relu,
                        float* y, void* stream);
/* The same net's FIRST convolution (4 input channels: the frame stack; w[32][3][3][4]): forward y[B, H, W, 32] = act(b + W x)
 * (relu != 0: ReLU) and weight / bias gradient (slabs part[slabs][32][36], dbp[slabs][32], slabs as above) straight from the
 * 4-channel rows -- no [B H W, 36] column matrix. -2 for W > 128. */
int ia_conv3x3_c4_forward(const float* x, const float* w, const float* bias, int B, int H, int W, int relu, float* y, void* stream);
int ia_conv3x3_c4_wgrad(const float* dz, const float* x, int B, int H, int W, float* part, float* dbp, void* stream);
/* Backward of "ReLU, then AdaptiveAvgPool2d(1)" on channel-last y[B, HW, C] in one pass: dz = y > 0 ? dout[b, c] / HW : 0
 * (C % 4 == 0; else -2). */
int ia_avgpool_relu_backward(const float* dout, const float* y, int B, int HW, int C, float* dz, void* stream);
/* ia_gemm_f32 with the [rows, KH*KW*Cin] operand of a convolution given implicitly as the im2col view of the
 * channel-last activation tensor x[Bn, H, W, Cin] (`torch.nn.Conv2d` forward / weight gradient without a column
 * buffer; [SB3 torch_layers.NatureCNN] cnn.2 / cnn.4). mode 0 (NT): A = x, C[M = Bn*OH*OW, N = Cout] =
 * act(view . B[N, K]^T + bias); mode 2 (TN): A = dout[rows, M = Cout], B = x, C_s[M, N = KH*KW*Cin] per K split
 * (+ dbias = column sums of dout). Needs Cin % 4 == 0 and (KW*Cin) % 32 == 0. */
int ia_gemm_f32_im2col(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                       int K, const float* bias, int act, int splits, float* dbias, int H, int W, int Cin, int KH, int KW,
                       int S, void* stream);

/* The same view with zero padding P on every side (taps outside the image read as 0) and, for mode 0, an optional
 * scatter of the output rows: cmap = {S_out, py, px, H_out, W_out} (HOST pointer to 5 ints) writes row (b, y', x') of
 * the [OH, OW] output grid to row (b, y'*S_out + py, x'*S_out + px) of a [H_out, W_out] grid in C; py < 0: all
 * S_out^2 classes in one GEMM (they share the view): N = S_out^2 * Cc, column class*Cc + c goes to channel c of the
 * class's row, C has Cc columns. With these the
 * INPUT gradient of a convolution is again an implicit GEMM over dout (`torch.nn.Conv2d` backward w.r.t. its input):
 * one padded stride-1 convolution with the flipped / transposed weights per sub-pixel class (py, px) of the stride --
 * no `dcol` buffer, no col2im; `relu_mask` (nullable, laid out like C) zeroes the outputs where the layer below's ReLU
 * output is <= 0. Also serves padded ("same") convolutions (`util/networks.py:286-357` build_cnn). */
int ia_gemm_f32_im2col_pad(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N,
                           int K, const float* bias, int act, int splits, float* dbias, int H, int W, int Cin, int KH,
                           int KW, int S, int P, const int* cmap, const float* relu_mask, void* stream);

/* ---- NatureCNN's first layer as an implicit GEMM (csrc/conv1_implicit.hip): Conv2d(4, 32, 8, stride 4) on uint8
 * [B, 4, H, W] frames with x * scale folded in ([SB3 torch_layers.NatureCNN] cnn.0 + [SB3 preprocess_obs]); no column
 * buffer. `ia_conv1_u8_implicit_ok`: 1 when the shape is covered (4 channels, 8x8 / 4, 32 filters, W % 4 == 0, image
 * fits the LDS budget), else use ia_im2col_u8_nchw + ia_gemm_f32. forward: out[B*OH*OW, 32] = relu(conv + bias),
 * ws = 8192 floats. wgrad: dW[32, 256] / db[32] (=, or += with `accumulate`) from dout[B*OH*OW, 32] (already masked
 * by the ReLU), ws = ia_conv1_u8_wgrad_ws_floats(B) floats; per-workgroup partials are summed in workgroup order. */
int ia_conv1_u8_implicit_ok(int C, int H, int W, int KH, int KW, int S, int Cout);
int ia_conv1_u8_forward(const uint8_t* x, int B, int H, int W, const float* weight, const float* bias, float scale,
                        float* ws, float* out, void* stream);
long long ia_conv1_u8_wgrad_ws_floats(int B);
int ia_conv1_u8_wgrad(const uint8_t* x, int B, int H, int W, const float* dout, float scale, float* ws, int accumulate,
                      float* dW, float* db, void* stream);
/* Categorical head ([SB3 CategoricalDistribution] log_prob / entropy of torch.distributions.Categorical):
 * logp[r] = log_softmax(logits[r])[action r], entropy[r]; dlogits (nullable) = gradient of
 * logp_coef*logp + ent_coef*entropy per row (BC: -share/B and -ent_weight*share/B, bc.py:138-156,494-499). */
int ia_categorical_loss(const float* logits, int ldl, const float* actions, int B, int A, float logp_coef,
                        float ent_coef, float* logp, float* entropy, float* dlogits, void* stream);

/* ---- actor-critic heads + PPO loss for towers of ANY shape (csrc/ppo_general.hip) -------------------------------
 * Policies whose `net_arch` the fused kernels above do not cover run their towers on ia_mlp_forward/_backward and
 * take the pieces between them from here ([SB3 distributions.py] DiagGaussianDistribution / CategoricalDistribution,
 * [SB3 ppo.py] PPO.train loss; reference call sites: adversarial/common.py:490-496, data/rollout.py:288-379).
 * ia_gauss_act: actions = mean + exp(log_std) * noise, clipped to [low, high] (clipped/logp nullable),
 * logp = Normal(mean, exp(log_std)).log_prob(actions).sum(-1). ia_gauss_eval: logp / entropy of given actions. */
int ia_gauss_act(const float* mean, const float* log_std, const float* noise, const float* low, const float* high, int n,
                 int A, float* actions, float* clipped, float* logp, void* stream);
int ia_gauss_eval(const float* mean, const float* log_std, const float* actions, int n, int A, float* logp, float* entropy,
                  void* stream);
/* out2 = (mean, unbiased std) of x[n] -- the minibatch advantage normalisation of PPO.train; one block, fixed order */
int ia_adv_moments(const float* x, int n, float* out2, void* stream);
/* torch.nn.utils.clip_grad_norm_ on one flat gradient (in place); norm_out (nullable) receives the total norm.
 * ws (nullable): ia_clip_grad_norm_ws_floats() floats -- with it, gradients of >= 65 536 entries are reduced by a grid of
 * blocks (partials folded in a fixed order) instead of one block. */
long long ia_clip_grad_norm_ws_floats(void);
int ia_clip_grad_norm(float* grad, long long n, float max_norm, float* norm_out, float* ws, void* stream);
/* One PPO minibatch at the heads: `out` = Gaussian means or Categorical logits [B, A] (A <= 64), `actions` [B, A]
 * (Box) or [B] fp32 indices (Discrete), adv_ms = ia_adv_moments of `adv` or NULL (no normalisation).
 * Writes d loss / d out, d loss / d values, d loss / d log_std (Box) and stats[8] = {policy_gradient_loss,
 * value_loss, entropy_loss, approx_kl, clip_fraction, loss, 0, 0}; ws: ia_ppo_head_loss_ws_floats(B) floats. */
long long ia_ppo_head_loss_ws_floats(int B);
int ia_ppo_head_loss(int discrete, const float* out, const float* log_std, const float* values, const float* actions,
                     const float* old_logp, const float* adv, const float* ret, const float* adv_ms, int B, int A,
                     float clip_range, float ent_coef, float vf_coef, float* d_out, float* d_values, float* dlog_std,
                     float* ws, float* stats, void* stream);

/* One PPO epoch = consecutive minibatches of the device-resident permutation `perm[T*n_envs]`
 * (host-drawn np.random.permutation); stats is [n_minibatches][8] or NULL. */
int ia_ppo_epoch(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                 int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                 const float* advantages, const float* returns, const int64_t* perm, int T, int n_envs,
                 int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                 float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                 float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream);
/* `n_epochs` consecutive epochs (the `for epoch in range(n_epochs)` loop of SB3's PPO.train) in one call: `perms` =
 * the epochs' permutations back to back [n_epochs][T*n_envs], stats [n_epochs][n_minibatches][8] or NULL,
 * ws: ia_ppo_ws_floats(d, batch, n_epochs * T * n_envs). The epochs run as ONE sequence of minibatches (gather and
 * statistics of every epoch in one launch each, the epoch kernels back to back): needs T*n_envs % batch_size == 0 --
 * IA_ERR_UNSUPPORTED otherwise, nothing launched (call ia_ppo_epoch per epoch). Same values as n_epochs ia_ppo_epoch calls. */
int ia_ppo_epochs(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                 int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                 const float* advantages, const float* returns, const int64_t* perms, int n_epochs, int T, int n_envs,
                 int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                 float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                 float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream);

/* A whole [SB3 PPO.train]: n_epochs passes over consecutive minibatches of perm[n_epochs][T*n_envs]
 * in ONE persistent launch (hidden = 32): nblk gradient blocks keep the parameters in LDS and
 * Adam's moments in registers, meet at one grid barrier per optimiser step, reduce the gradient
 * slabs redundantly in fixed order and apply identical clip + Adam updates; a further block runs
 * ahead computing each minibatch's advantage statistics and train-mode RunningNorm update.
 * ia_ppo_update_ws_floats returns 0 when the shape is not covered (use ia_ppo_epoch per epoch);
 * ws must be zeroed once by the caller, word 8 of it is a sticky error flag (a bounded grid wait
 * timed out: results invalid). stats: [n_epochs*n_minibatches][8] or NULL.
 * `obs` is read in 16-byte pieces, a row's last piece up to 12 bytes past the row when obs_dim % 4 != 0: the ALLOCATION behind
 * `obs` must extend at least 12 bytes past row T*n_envs - 1 (SB3's rollout tile has T + 1 time slices -- the observation behind
 * the last step -- and satisfies this by construction; a tile of exactly T slices at the end of a mapped region faults).
 * ia_ppo_update_xcd_pack(1) (several gradient workgroups that fit one XCD's 32 CUs) launches 8x the blocks so that the
 * working ones share one XCD; the gradient workgroups then check their placement against each other (XCC ids, one word
 * exchange per launch) and, when they do sit on one XCD, store the step's (value, sequence) words at workgroup scope --
 * L2-resident instead of one fabric write per word. Default 0 (spread over all XCDs): packed measured +1-2 % at config P
 * on one box, nothing on another, and -3 ... -5 % where the row gathers dominate (1 000-step rollouts, Ant-width rows). */
int64_t ia_ppo_update_ws_floats(const ia_policy_desc* d, int batch_size);
int ia_ppo_update_xcd_pack(int on);
/* ia_ppo_update returns IA_ERR_UNSUPPORTED (-2) without launching when its workgroups (which meet at grid
 * barriers) would not all be resident at once: occupancy query x compute units < grid. Tests: pretend the
 * device has n compute units (0 = ask the device). */
int ia_ppo_update_assume_cus(int n);
int ia_ppo_update(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                  int32_t* norm_count, int update_norm, const float* obs, const float* actions, const float* old_logp,
                  const float* advantages, const float* returns, const int64_t* perm, int n_epochs, int T, int n_envs,
                  int batch_size, int normalize_adv, float clip_range, float ent_coef, float vf_coef,
                  float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                  float adam_eps, int64_t adam_steps_done, float* ws, float* stats, void* stream);

/* Row-sharded data-parallel [SB3 PPO.train] (no reference counterpart: the reference is single-process, SURVEY section 5
 * row "Distributed"; the partitioning is SURVEY 8e (1)-(3) / BASELINE.json north_star "sharded by env-batch ... all-reduce
 * of ... policy grads"). One process per GPU; `obs` ... `returns` are the ALL-GATHERED rollout tile [T, n_envs] with
 * n_envs = world x the rank's environments, `perm` the permutations every rank shares; each optimiser step works on a
 * global minibatch of world x rows_per_rank rows of which this rank's workgroups take rows [rank * rows_per_rank, ...).
 * The per-step exchange -- one record per rank: partial gradient + loss-statistic sums -- happens INSIDE the persistent
 * kernel through peer-mapped device memory (xGMI between GPUs): every rank writes its record into every rank's receive
 * area as 8-byte (value, step sequence number) words with system-scope stores -- an aligned 8-byte store arrives whole, so
 * no flag, acknowledgement or fence is needed: the receiver spins on the words themselves --; all workgroups sum the
 * records in rank order (bit-identical replicas), clip by the global norm (torch `clip_grad_norm_` after the sum) and
 * apply Adam. Advantage statistics and the train-mode feature RunningNorm of a minibatch are computed over the global
 * minibatch on every rank. `recv` (ia_ppo_shard_recv_bytes): this rank's area, zeroed ONCE at allocation; `peer_recv[r]`:
 * rank r's area as mapped into this process (own rank included); `seq_base`: optimiser steps exchanged through these
 * areas so far, identical on every rank, only ever growing. A rank that waits longer than
 * `timeout_s` for a peer's record sets the sticky error word of `ws` (word 8: 2) and leaves. `loopback` != 0 (cost model
 * on one process, tools/dp_overhead.py): all peer areas are the caller's own and it writes its record once per source rank. */
int64_t ia_ppo_update_sharded_ws_floats(const ia_policy_desc* d, int rows_per_rank, int world);
int64_t ia_ppo_shard_recv_bytes(const ia_policy_desc* d, int world);
int ia_ppo_update_sharded(const ia_policy_desc* d, float* params, float* params_t, float* norm_mean, float* norm_var,
                          int32_t* norm_count, int update_norm, const float* obs, const float* actions,
                          const float* old_logp, const float* advantages, const float* returns, const int64_t* perm,
                          int n_epochs, int T, int n_envs, int rows_per_rank, int normalize_adv, float clip_range,
                          float ent_coef, float vf_coef, float max_grad_norm, float* exp_avg, float* exp_avg_sq, double lr,
                          double beta1, double beta2, float adam_eps, int64_t adam_steps_done, float* ws, float* stats,
                          int world, int rank, uint32_t seq_base, void* recv, void* const* peer_recv, int loopback,
                          double timeout_s, void* stream);
/* Peer-mapped device memory for the exchange above (the only entries that allocate, map, free or synchronise):
 * ia_peer_alloc: `bytes` of zeroed device memory, fine-grained (system-scope coherent) when the runtime grants it
 * (*fine_grained says which); ia_peer_ipc_export / _open / _close: the 64-byte hipIpc handle of a block and its mapping in
 * another process (the ranks exchange handles over torch.distributed); ia_peer_handshake: one-wave kernel that writes
 * `token` (non-zero, new for every call) into word [rank] of every rank's `peer_words[r]` and waits `timeout_s` for all
 * `world` words of `own_words` to carry it -- *result = 1 when the mapping, peer writes and system-scope polling all work
 * (the trainer falls back to the replicated update otherwise). */
int ia_peer_alloc(size_t bytes, void** out, int* fine_grained);
int ia_peer_free(void* p);
int ia_peer_ipc_export(void* p, unsigned char* handle64);
int ia_peer_ipc_open(const unsigned char* handle64, void** out);
int ia_peer_ipc_close(void* p);
int ia_peer_handshake(int world, int rank, uint32_t token, uint32_t* own_words, uint32_t* const* peer_words,
                      double timeout_s, int* result, void* stream);

#ifdef __cplusplus
}
#endif
#endif
