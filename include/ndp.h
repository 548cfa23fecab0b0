/* ndp.h -- C ABI of libndp_hip.so: the MI355X (gfx950) implementation of the
 * GAN-training hot path of goodmattg/ndivplanning.
 *
 * The reference has no FFI layer; its boundary for this path is Python
 * (models/gan.py, diversity.py, train_gan.py).  The functions below are what a
 * binding for that path calls (ctypes stub: INTEGRATION.md).  Each one names the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions (all functions):
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 unless
 *     stated otherwise; the caller owns every buffer, nothing is allocated or
 *     freed inside, no call synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); work
 *     is enqueued on it and the call returns immediately (graph-capturable);
 *   - return value 0 = enqueued; non-zero = NDP_E_* and nothing was launched;
 *     ndp_last_error() returns a thread-local message for the last failure;
 *   - no global mutable state: calls from different host threads on different
 *     streams are independent (autograd invokes backward from its own thread).
 *
 * Parameter vectors: the networks' parameters are ONE flat fp32 vector each, in
 * state_dict order fc1.weight, fc1.bias, fc2.weight, ... (weight [out][in]
 * row-major, as nn.Linear stores it).  Gradients and Adam moments use the same
 * layout.  ndp_g_param_count / ndp_d_param_count give the lengths.
 */
#ifndef NDP_H_
#define NDP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDP_VERSION 135          /* 0.3.2: ndp_fm_* (forward / next-frame model), ndp_fm_backward, ndp_fm_side_stream */

#define NDP_OK            0
#define NDP_E_ARG         1      /* bad argument (shape, alignment, null) */
#define NDP_E_LAUNCH      2      /* hipLaunchKernel / hipMemsetAsync failed */
#define NDP_E_UNSUPPORTED 3      /* valid request this build does not cover */

#define NDP_CODE_DIM      256    /* cat(state_code, target_code): train_gan.py:155 */
#define NDP_ACTION_DIM    4      /* models/gan.py:71,94 */
#define NDP_MAX_NOISE_DIM 16
#define NDP_MAX_SAMPLES   256    /* K, diversity samples per row */
#define NDP_ROW_PAD       32     /* row counts of workspaces are padded to this */

int         ndp_version(void);
const char *ndp_last_error(void);

/* Number of fp32 parameters of Decoder(noise_dim) (models/gan.py:61-71) and of
 * Discriminator() (models/gan.py:89-97): 83,780 at noise_dim 2 and 58,305. */
int64_t ndp_g_param_count(int noise_dim);
int64_t ndp_d_param_count(void);

/* rows rounded up to NDP_ROW_PAD */
int64_t ndp_pad_rows(int64_t rows);

/* ------------------------------------------------------------------ NDiv ---
 * diversity.compute_pairwise_divergence(recodes=x, codes=z), forward and
 * backward in one pass (diversity.py:8-19, 36-41):
 *   loss = sum_{n,i,j} relu(0.8 * dz_ij/sum_j dz_ij - dx_ij/sum_j dx_ij)
 * with the row sums treated as constants in the gradient (diversity.py:18) and
 * a zero sub-gradient where dx_ij == 0 (torch.norm's backward).
 *   x        [n, k, cx]   recodes (generated actions)
 *   z        [n, k, cz]   codes (noise)
 *   loss_out [1]          written (not accumulated)
 *   grad_x   [n, k, cx]   d(grad_scale*loss)/dx, written; may be NULL
 *   partials [ndp_ndiv_partials(n,k)] scratch
 * 1 <= k <= NDP_MAX_SAMPLES, 1 <= cx,cz <= 16.  k == 1 yields NaN like the
 * reference (0/0). */
int64_t ndp_ndiv_partials(int64_t n, int k);
int ndp_ndiv_fwd_bwd(const float *x, int cx, const float *z, int cz, int64_t n, int k,
                     float grad_scale, float *loss_out, float *grad_x, float *partials,
                     void *stream);

/* ---------------------------------------------------------- Generator G ---
 * Decoder.forward (models/gan.py:79-86): action_hat = fc5(relu(fc4(relu(fc3(
 * relu(fc2(relu(fc1(cat[code, noise]))))))))).
 * The input is given as its two parts so that the K-fold repeat of the code
 * (train_gan.py:42-47) never has to be materialised:
 *   code   row r of the network input uses code[(r / code_rep) * ld_code ...+256)
 *   noise  row r uses noise[r * ld_noise ...+noise_dim)
 * For a plain z [m, 256+nz] pass code=z, ld_code=256+nz, code_rep=1,
 * noise=z+256, ld_noise=256+nz.
 *   acts   NULL, or workspace of ndp_g_acts_floats(m) floats receiving the
 *          hidden activations h1..h4 (needed by ndp_g_backward)
 *   action_hat [m, 4] */
int64_t ndp_g_acts_floats(int64_t m);
int ndp_g_forward(const float *g_params, int noise_dim,
                  const float *code, int64_t ld_code, int code_rep,
                  const float *noise, int64_t ld_noise, int64_t m,
                  float *acts, float *action_hat, void *stream);

/* Backward of Decoder.forward for the parameters (what autograd computes at
 * train_gan.py:202 for the Decoder): given d_action [m,4] = dLoss/d action_hat
 * and the activations saved by ndp_g_forward, writes
 *   grad [ndp_g_param_count]  dLoss/d params (written, not accumulated)
 *   ws   scratch of ndp_g_bwd_ws_floats(m, noise_dim) floats
 * No gradient is produced for the network input (the reference detaches the
 * codes and never differentiates the noise: train_gan.py:152-153, 44). */
int64_t ndp_g_bwd_ws_floats(int64_t m, int noise_dim);
int ndp_g_backward(const float *g_params, int noise_dim,
                   const float *code, int64_t ld_code, int code_rep,
                   const float *noise, int64_t ld_noise, int64_t m,
                   const float *acts, const float *d_action,
                   float *grad, float *ws, void *stream);

/* ------------------------------------------------------ Discriminator D ---
 * Discriminator.forward (models/gan.py:104-110): logits = fc4(lrelu(fc3(lrelu(
 * fc2(lrelu(fc1(cat[action, code]))))))), slope 0.01.
 *   action row r uses action[(r / action_rep) * 4 ...+4)
 *   code   row r uses code[(r / code_rep) * ld_code ...+256)
 *   logits [m] */
int ndp_d_forward(const float *d_params,
                  const float *action, int action_rep,
                  const float *code, int64_t ld_code, int code_rep, int64_t m,
                  float *logits, void *stream);

/* Backward of Discriminator.forward given d_logits [m] = dLoss/d logits
 * (recomputes the forward inside the kernel; nothing needs to be saved):
 *   grad     [ndp_d_param_count] or NULL   dLoss/d params (written)
 *   d_action [m,4] or NULL                 dLoss/d action (written; needs action_rep==1)
 *   ws       scratch of ndp_d_bwd_ws_floats(m) floats (only used when grad != NULL) */
int64_t ndp_d_bwd_ws_floats(int64_t m);
int ndp_d_backward(const float *d_params,
                   const float *action, int action_rep,
                   const float *code, int64_t ld_code, int code_rep, int64_t m,
                   const float *d_logits, float *grad, float *d_action,
                   float *ws, void *stream);

/* ----------------------------------------------------------------- Adam ---
 * torch.optim.Adam.step for one flat parameter vector (train_gan.py:98-104,
 * 184, 203): m += (1-b1)(g-m); v = b2 v + (1-b2) g^2;
 * p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),  t = step_count[0] + 1.
 * step_count is a DEVICE int32[4] "Adam state word": [0] = number of updates applied
 * so far (the call increments it), [1..3] = scratch owned by the library (the bias
 * corrections of the current update, computed once on the device in fp64), so that a
 * captured graph replays with the right bias correction. */
int ndp_adam_step(float *params, const float *grad, float *exp_avg, float *exp_avg_sq,
                  int64_t n, int32_t *step_count, float lr, float beta1, float beta2,
                  float eps, void *stream);

/* ------------------------------------------------------ fused train step ---
 * One iteration of the train_gan.py loop body (train_gan.py:159-203) on a local
 * shard of `flat` rows (FLAT = batch*(traj_len-1)), each with K = num_sample
 * diversity samples, M = flat*K rows through G and D.  The step is split where
 * a data-parallel driver has to all-reduce gradients; single-GPU training sets
 * fuse_adam = 1 and never sees the gradients.
 *
 *   phase A  ndp_step_d_grads : [first call of the step: G forward]
 *            D(real), D(fake) forward, BCE, D backward  -> D gradient
 *            (train_gan.py:165-183); with fuse_adam the D Adam update too (184).
 *            `actions` / `codes` hold ONE row per FLAT row; the reference feeds D
 *            K = num_sample identical copies of each (repeat_interleave,
 *            train_gan.py:140-156), so on the first call of a step the real pass
 *            runs on the FLAT distinct rows with every row's BCE term and loss
 *            gradient weighted K: the same sums, 1/K of the rows.  Repeat calls
 *            (run_g_forward = 0) run both passes on all M rows.
 *   phase B  ndp_step_g_grads : D(fake) forward with the UPDATED D, G loss,
 *            NDiv loss + gradient, backward through D and G -> G gradient
 *            (train_gan.py:187-202); with fuse_adam the G Adam update too (203)
 *   ndp_adam_step              for the non-fused (data-parallel) case
 *
 * Scaling for data parallelism: BCE terms are means over
 * the GLOBAL row count, so inv_m_global = 1/(M summed over ranks); the NDiv term
 * is a sum and is not scaled.  Gradients of all ranks are then SUMMED.
 */
struct ndp_p2p;
typedef struct ndp_step_config {
  int32_t noise_dim;           /* training.gan.noise_dim  (1..16) */
  int32_t num_sample;          /* training.gan.num_sample (1..256) */
  int64_t flat;                /* local FLAT rows */
  float   inv_m_global;        /* 1 / global M */
  float   pairwise_div_factor; /* training.gan.pairwise_div_factor */
  float   lr, beta1, beta2, eps;
  int32_t fuse_adam;           /* 1: apply Adam inside the phase; 0: leave grads */
  int32_t device_noise;        /* 1: G forward draws the noise itself (see `noise` below) */
  uint64_t noise_seed;         /* stream id of the device noise (e.g. the rank) */
  const struct ndp_p2p *p2p;   /* NULL, or the peer-to-peer gradient exchange (see below): the
                                * kernel that sums the split-K slabs then also SUMS the gradient
                                * over ranks before Adam, so a data-parallel step has the same
                                * launches as a single-GPU step and is graph-capturable */
} ndp_step_config;

/* Caller-owned persistent state of one trainer (all device memory). */
typedef struct ndp_step_buffers {
  float   *g_params, *g_grad, *g_exp_avg, *g_exp_avg_sq;   /* [ndp_g_param_count] */
  float   *d_params, *d_grad, *d_exp_avg, *d_exp_avg_sq;   /* [ndp_d_param_count] */
  int32_t *g_step, *d_step;                                 /* Adam state words, int32[4] each */
  float   *losses;       /* [4]: D_loss, G_loss, pair_div (local shares), unused */
  float   *loss_sums;    /* [4]: running sums of the above (epoch averages) or NULL */
  float   *action_hat;   /* [pad(M), 4] generated actions of the current step */
  float   *workspace;    /* ndp_step_workspace_floats(cfg) floats */
} ndp_step_buffers;

int64_t ndp_step_workspace_floats(const ndp_step_config *cfg);

/* codes [flat,256], actions [flat,4] (ground truth), noise [flat,K,nz].
 * run_g_forward: 1 on the first D step of an iteration, 0 on repeats
 * (discrim_steps_per_gen > 1 re-uses action_hat, train_gan.py:172).
 * With cfg->device_noise the G forward fills `noise` itself with U[0,1) drawn from the
 * counter-based stream (noise_seed, offset = g_step[0]) -- the torch.FloatTensor(...)
 * .uniform_() of diverse_sampling (train_gan.py:44) without the host round trip -- and
 * the later kernels read it from there; otherwise `noise` is an input. */
int ndp_step_d_grads(const ndp_step_config *cfg, const ndp_step_buffers *buf,
                     const float *codes, const float *actions, float *noise,
                     int run_g_forward, void *stream);
int ndp_step_g_grads(const ndp_step_config *cfg, const ndp_step_buffers *buf,
                     const float *codes, const float *actions, const float *noise,
                     void *stream);

/* The step kernels read the layers' weights from lane-ordered ("packed") copies kept in the
 * workspace; the fused Adam updates refresh them.  Call ndp_step_pack_params once before
 * the first step and again whenever g_params / d_params were written by anything else
 * (checkpoint load, a host-side optimizer, ...). */
int ndp_step_pack_params(const ndp_step_config *cfg, const ndp_step_buffers *buf, void *stream);

/* Non-fused (data-parallel) update: Adam for network `which` (0 = D, 1 = G) from
 * buf->d_grad / buf->g_grad (after the all-reduce), refreshing the packed copies.  Must follow
 * the ndp_step_d_grads / ndp_step_g_grads call that produced the gradient: that call also
 * advanced the network's Adam state word (step count and bias corrections). */
int ndp_step_apply_adam(const ndp_step_config *cfg, const ndp_step_buffers *buf, int which,
                        void *stream);

/* The same device noise stream as a stand-alone call: out[i] = U[0,1) from
 * Philox-4x32-10 keyed by seed, counter (i/4, *offset_dev), word i%4. */
int ndp_uniform_noise(float *out, int64_t n, uint64_t seed, const int32_t *offset_dev,
                      void *stream);

/* ------------------------------------------- peer-to-peer gradient exchange ---
 * The data-parallel exchange of SURVEY.md section 8e (two flat SUM all-reduces
 * per step: 58,305 and 83,780 floats) without a collective library: the
 * reference has no collectives (single process, train_gan.py:115-207); this is
 * what replaces "one optimizer sees the whole batch" when the batch is sharded
 * over one process per GPU.
 *
 * Every rank owns one REGION of device memory (uncached / fine-grained, so that
 * stores arriving over xGMI and the polling loads bypass the caches), exported
 * with hipIpc and mapped by every peer.  One exchange, inside the kernel that
 * has just summed the rank's split-K slabs:
 *   push   each workgroup stores its 256 gradient values into the inbox slot
 *          [src = this rank][step parity] of EVERY peer's region, waits for the
 *          stores to be acknowledged, then sets flag[src][workgroup] = step there
 *          (all accesses are system-scope atomics on uncached memory: no cache
 *          maintenance, no fences);
 *   wait   it polls the flags the peers set in its OWN region (bounded by
 *          timeout_ms: on expiry the region's status word is set, the wait is
 *          skipped from then on and ndp_p2p_status reports it -- never a hang);
 *   sum    own value + the peers' values from its own inbox in rank order
 *          0..world-1: every rank computes bit-identical sums, so the replicas
 *          stay bit-identical, run after run.
 * Inboxes are double-buffered on the step's parity: a peer can be at most one
 * exchange ahead (it needs this rank's next push to go further), so no second
 * barrier is needed.  `step` is the network's Adam step count (state word [0]),
 * which must advance by one per exchange and be the same on all ranks.
 * These are the only functions of the library that allocate or synchronise. */
#define NDP_P2P_MAX_RANKS    8
#define NDP_P2P_HANDLE_BYTES 64          /* sizeof(hipIpcMemHandle_t) */

typedef struct ndp_p2p {
  int32_t world, rank;
  int32_t timeout_ms;                    /* bound of one wait (0 = 10,000) */
  int32_t reserved;
  void   *region[NDP_P2P_MAX_RANKS];     /* region[rank] = own allocation, others = mapped peers */
} ndp_p2p;

int64_t ndp_p2p_region_bytes(void);
int ndp_p2p_region_alloc(void **region);                 /* zero-filled; synchronises */
int ndp_p2p_region_free(void *region);
int ndp_p2p_region_reset(void *region);                  /* zero flags + status; synchronises */
int ndp_p2p_export(void *region, void *handle_out);      /* NDP_P2P_HANDLE_BYTES bytes, host memory */
int ndp_p2p_open(const void *handle, void **mapped_out); /* a PEER process's handle */
int ndp_p2p_close(void *mapped);
/* status word of the own region: 0 = ok, 1 + r = a wait for rank r timed out (HOST int out) */
int ndp_p2p_status(const ndp_p2p *p2p, int32_t *status_out);
/* What the first waiter that gave up was waiting for (HOST int32[NDP_P2P_DIAG_WORDS] out; synchronises):
 * {status code, workgroup, net, expected step, flag value it saw, peer rank, 100 MHz ticks waited, 0}. */
#define NDP_P2P_DIAG_WORDS 8
int ndp_p2p_diagnostics(const ndp_p2p *p2p, int32_t *words_out);
/* Copy the status word to PINNED host memory behind everything already enqueued on `stream`, without
 * synchronising: a training loop polls the value of the previous launch for free and aborts early. */
int ndp_p2p_status_async(const ndp_p2p *p2p, int32_t *pinned_host_out, void *stream);
/* PCI bus id ("0000:c1:00.0") of the CURRENT device into out[len >= 16]: ranks that report the same id
 * share a GPU, and the in-kernel exchange needs every rank's reduce kernel co-resident -- which only a
 * GPU per rank guarantees (ndivplanning_amd/dp.py refuses the exchange for such ranks unless forced). */
int ndp_device_pci_bus_id(char *out, int len);
/* The exchange alone: out[i] = sum over ranks of in[i] (n <= the capacity of `net`'s inbox:
 * ndp_d_param_count() for net 0, ndp_g_param_count(NDP_MAX_NOISE_DIM) for net 1).  step_word:
 * device int32, same value on every rank, larger than at the previous exchange on this net. */
int ndp_p2p_all_reduce(const ndp_p2p *p2p, int net, const float *in, float *out, int64_t n,
                       const int32_t *step_word, void *stream);

/* ------------------------------------------------------------ image encoder ---
 * models.image_autoencoder.Encoder.forward in eval mode without gradient
 * (image_autoencoder.py:35-49; train_gan.py:75-76 loads it, 152-153 calls it under
 * .detach()): images [n,3,128,128] (NCHW, as the reference's loader delivers them) ->
 * codes [n,128].  All six layers are implicit GEMMs on the fp32 matrix pipe (conv1: K = 27
 * padded to one 32-wide step) over NHWC activations kept in `workspace`.
 * packed_params: ndp_encoder_param_floats() floats, BatchNorm (eval: running statistics,
 * eps 1e-5) of conv1..conv3 folded into weights and biases:
 *   conv1  w[27][64] with k = ci*9 + kh*3 + kw, then bias[64];
 *   conv2..conv6  w[Cout][KH][KW][Cin], then bias[Cout]   (in this order, back to back).
 * Images are processed in passes of at most 512; workspace: ndp_encoder_workspace_floats(n). */
int64_t ndp_encoder_param_floats(void);
int64_t ndp_encoder_workspace_floats(int64_t n_images);
int ndp_encoder_forward(const float *packed_params, const float *images, int64_t n_images,
                        float *codes, float *workspace, void *stream);
/* The same from DECODED CAMERA FRAMES: frames_hwc [n][128][128][3] bytes (what PIL's JPEG decoder hands the
 * reference's loader, utils/trajectory_loader.py:48-56).  The reference turns them into its [-1, 1] float tensors on
 * the host -- utils/hdf5_load.py:9-11: (ToTensor()(image) - 0.5) * 2.0, ToTensor = byte -> float32, / 255 -- permutes to
 * CHW and uploads 4 bytes per value (train_gan.py:119-124).  Here the first convolution gathers from the bytes and
 * applies the same three fp32 operations (a 256-entry table): bit-identical codes from a quarter of the upload. */
int ndp_encoder_forward_u8(const float *packed_params, const uint8_t *frames_hwc, int64_t n_images,
                           float *codes, float *workspace, void *stream);

/* ------------------------------------------- forward (next-frame) model ---
 * models.forward_encoder.ForwardAutoencoder (forward_encoder.py:20-114) and one iteration
 * of its training loop (train_forward_model.py:98-112): U-Net of stride-2 convolutions /
 * transposed convolutions with BatchNorm, conditioned on the 4-d action.
 *   ndp_fm_forward      replaces  forward_autoencoder(state_cur, action)   (forward_encoder.py:105-114)
 *                       training != 0: batch-statistics BatchNorm, returns the residual, moves the running
 *                       statistics when running_stats != NULL; training == 0: running statistics,
 *                       returns state_cur + residual (what control_evaluation.py / mpc_eval.py call)
 *   ndp_fm_train_grads  replaces  loss = mse(model(cur, a), fut - cur); zero_grad(); loss.backward()
 *                       (train_forward_model.py:102-109): loss[0] = the MSE, *loss_sum += it (NULL: not kept),
 *                       grad = every gradient, resid_out (NULL or [n,3,128,128]) = the prediction
 *   ndp_fm_backward     replaces  loss.backward() for ANY loss of the residual: d_resid [n,3,128,128] = d loss / d
 *                       residual; must follow ndp_fm_forward(training != 0) on the same n images with the same workspace
 *                       and nothing in between (the activations live there); grad = every gradient
 *   ndp_fm_apply_adam   replaces  optimizer.step() (:110) for the flat parameter vector, and rebuilds the
 *                       second weight order in the workspace (step_count: the Adam state word of ndp_adam_step)
 * Images [n,3,128,128] NCHW and actions [n,4] as the reference's loader delivers them.
 * Parameters are ONE flat fp32 vector (ndp_fm_param_floats() floats), gradients and Adam moments have the same
 * layout: per layer (conv1..6, deconv1..6, conv_refine_1, conv_refine_2) the weight then the bias, then per
 * BatchNorm in use (conv1..3_bn, deconv1..6_bn, conv_refine_1_bn) weight then bias.  Weights are stored
 * tap-major with the channel the kernels read along innermost, zero-padded:
 *   Conv2d           [cout_pad][kh][kw][cin_pad]   = weight.permute(0, 2, 3, 1)
 *   ConvTranspose2d  [cin_pad][kh][kw][cout_pad]   = weight.permute(0, 2, 3, 1)
 * ndp_fm_layout(what, index, &offset, dims) describes every tensor: what 0 weight / 1 bias of layer `index`
 * (dims = rows, taps, columns, kind 0 conv / 1 transposed, cin, cout), 2 / 3 BatchNorm weight / bias, 4 / 5
 * running mean / variance (offsets into the running_stats vector of ndp_fm_stat_floats() floats; dims[0] =
 * channels, dims[3] = the layer the BatchNorm follows).  Padded entries must be zero and stay zero under Adam.
 * The reference's conv4_bn / conv5_bn are never applied (forward_encoder.py:51-54) and are not part of the vector.
 * workspace: ndp_fm_workspace_floats(n) floats; its head holds the second weight order, which
 * ndp_fm_pack_params (after the parameters were written from outside) and ndp_fm_apply_adam rebuild -- the
 * same workspace pointer must be used for all calls.  ndp_fm_workspace_offset(n, t): where intermediate map t
 * (order of FmTensor in csrc/ndp_forward_model.inc) lives, for tests. */
int64_t ndp_fm_param_floats(void);
int64_t ndp_fm_stat_floats(void);
int64_t ndp_fm_workspace_floats(int64_t n_images);
int64_t ndp_fm_workspace_offset(int64_t n_images, int tensor);
int ndp_fm_layout(int what, int index, int64_t *offset, int64_t *dims /* [6] */);
int ndp_fm_pack_params(const float *params, float *workspace, void *stream);
int ndp_fm_forward(const float *params, float *running_stats, const float *state_cur,
                   const float *actions, int64_t n_images, int training, float *out,
                   float *workspace, void *stream);
int ndp_fm_train_grads(const float *params, float *running_stats, const float *state_cur,
                       const float *state_fut, const float *actions, int64_t n_images,
                       float *grad, float *loss, float *loss_sum, float *resid_out,
                       float *workspace, void *stream);
/* The weight gradients of the backward pass run on a stream of the library's own beside the caller's (fork / join by
 * events); ndp_fm_side_stream(0) keeps every launch on the caller's stream (per-kernel timing; stream capture: the fork
 * captures, but the HIP graph of it replayed at 3.9 ms against 2.0 ms eager -- and a graph of the single-stream step
 * gains nothing over eager either), returns the previous setting. */
/* ndp_fm_forward / ndp_fm_train_grads from byte frames [n][128][128][3] (see ndp_encoder_forward_u8): state_cur and
 * state_fut are normalised where they are read (the input gather and the loss), outputs are as above. */
int ndp_fm_forward_u8(const float *params, float *running_stats, const uint8_t *frames_cur,
                      const float *actions, int64_t n_images, int training, float *out,
                      float *workspace, void *stream);
int ndp_fm_train_grads_u8(const float *params, float *running_stats, const uint8_t *frames_cur,
                          const uint8_t *frames_fut, const float *actions, int64_t n_images,
                          float *grad, float *loss, float *loss_sum, float *resid_out,
                          float *workspace, void *stream);
int ndp_fm_side_stream(int on);
/* Cross-rank BatchNorm statistics for a data-parallel driver.  The per-channel sums a BatchNorm needs are accumulated
 * by the kernel that produces its input as 64-bit fixed-point integers (csrc/ndp_forward_model.inc, "epilogue
 * statistics").  With a function set here, every ndp_fm_forward(training) / ndp_fm_train_grads / ndp_fm_backward call
 * invokes fn(acc, words, stream, ctx) on the calling thread between the launch that fills an accumulator and the launch
 * that reads it; fn must enqueue, on `stream`, an in-place SUM over the ranks of the `words` int64 values at `acc` (device
 * memory inside the workspace) -- an RCCL all-reduce of ncclInt64.  Integer sums are exact and order-free: `world` ranks
 * with B / world images each then normalise, and update the running statistics, exactly as one process with B images
 * does (the reference trains on one device: train_forward_model.py:62); the BatchNorm weight / bias gradients stay each
 * rank's own share, for the gradient all-reduce to add up.  20 small collectives per iteration.  fn == NULL: off (per-rank
 * statistics, what torch's DistributedDataParallel does without SyncBatchNorm).  Process-wide. */
typedef void (*ndp_fm_stat_sync_fn)(void *acc, int64_t words, void *stream, void *ctx);
int ndp_fm_set_stat_sync(ndp_fm_stat_sync_fn fn, void *ctx, int world);

/* Gradient buckets for a data-parallel driver (the reference trains on one device: train_forward_model.py:62; the
 * north star asks for the all-reduce of the gradients "overlapped with backward").  ndp_fm_grad_buckets writes the 7
 * ranges (offset, count: floats of the flat gradient) in the order in which a backward pass completes them -- the weight
 * gradients come last layer first -- and returns 7.  Every ndp_fm_train_grads / ndp_fm_backward call records one event
 * per bucket where its last byte is written; ndp_fm_bucket_wait(b, stream) makes `stream` (the caller's communication
 * stream) wait for bucket b of the most recent such call on the current device, so that its all-reduce runs beside the
 * rest of the backward pass.  The buckets cover the whole vector exactly once. */
int ndp_fm_grad_buckets(int64_t *offsets, int64_t *counts, int capacity);
int ndp_fm_bucket_wait(int bucket, void *stream);
int ndp_fm_backward(const float *params, const float *d_resid, int64_t n_images, float *grad,
                    float *workspace, void *stream);
int ndp_fm_apply_adam(float *params, const float *grad, float *exp_avg, float *exp_avg_sq,
                      int32_t *step_count, float lr, float beta1, float beta2, float eps,
                      float *workspace, void *stream);

/* ------------------------------------------------------------ measurement ---
 * Per-kernel timing for bench.py: while enabled (per host thread) every kernel
 * this library launches is bracketed by hipEvents recorded on the stream it is
 * launched on.  Must be off during graph capture.
 * ndp_timing_collect synchronises on the recorded events, sums the elapsed ms
 * per kernel name and resets the record: names receives "name0;name1;..."
 * (at most names_len bytes), total_ms / counts up to max_kernels entries.
 * Returns the number of distinct kernels; 0 with ndp_last_error() set on failure. */
int ndp_timing_enable(int on);
int ndp_timing_collect(char *names, int names_len, float *total_ms, int32_t *counts,
                       int max_kernels);

#ifdef __cplusplus
}
#endif
#endif /* NDP_H_ */
