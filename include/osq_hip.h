/*
 * osq_hip.h -- C ABI of the MI355X (gfx950) fake-quant / observer hot path.
 *
 * The reference (wimh966/outlier_suppression) has no FFI: its hot path is a chain
 * of eager PyTorch ops inside quant_transformer/quantization/ (the .py files).  Each entry
 * point below replaces one such chain; the comment above it cites the reference
 * lines it stands in for.  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add to call them.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on
 *     that stream and never synchronises with the host;
 *   - scalars that the reference reads with `.item()` (scale, zero_point,
 *     min_val, max_val) stay on the device and are passed by pointer;
 *   - return value: 0 on success, a negative osq_status otherwise;
 *     osq_last_error() returns a thread-local description;
 *   - `workspace` is caller-owned device scratch of at least
 *     osq_workspace_bytes() bytes, zero-initialised ONCE by the caller and then
 *     reused (kernels leave it zeroed where they need it zeroed).  One
 *     workspace per stream in flight.
 */
#ifndef OSQ_HIP_H
#define OSQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* osq_stream;

/* Bumped whenever a signature or the workspace layout changes; the Python host refuses a library whose
 * osq_abi_version() differs from the number it was written against (a stale libosq_hip.so must be rebuilt).
 * 5: the LSQ / LSQ+ backward takes its summation order as an argument (`lanes` / `sum_lanes`). */
#define OSQ_ABI_VERSION 7

typedef enum osq_status {
    OSQ_OK = 0,
    OSQ_ERR_INVALID_ARGUMENT = -1,
    OSQ_ERR_HIP = -2,
    OSQ_ERR_UNSUPPORTED = -3
} osq_status;

/* zero-point storage: FixedFakeQuantize keeps int32 (fake_quant.py:105),
 * LSQPlusFakeQuantize an fp32 Parameter (fake_quant.py:174). */
typedef enum osq_zp_type { OSQ_ZP_INT32 = 0, OSQ_ZP_FLOAT32 = 1 } osq_zp_type;

/* how the learnable variants post-process (scale, zero_point) before quantising */
typedef enum osq_param_mode {
    OSQ_PARAM_FIXED = 0,   /* util_quant.py:11-26   use as is                                  */
    OSQ_PARAM_LSQ = 1,     /* util_quant.py:29-45   scale <- grad_scale(scale, g)              */
    OSQ_PARAM_LSQPLUS = 2, /* util_quant.py:48-67   zp <- round_ste(zp); both <- grad_scale()  */
    OSQ_PARAM_MODE_MASK = 3,
    /* flag for the per-tensor forward entry points, OR-ed into `mode`: first repair the parameters
     * IN PLACE the way LSQFakeQuantize / LSQPlusFakeQuantize.forward do while their observer is off
     * (fake_quant.py:152-153, 188-191: scale.abs_(); scale.clamp_(min=finfo(float32).eps); LSQ+ also
     * zero_point.clamp_(quant_min, quant_max)), then quantise with the repaired values -- the same
     * result as osq_lsq_sanitize followed by the plain call, in one launch.  These entry points therefore take
     * `scale` / `zero_point` without a const qualifier; without the flag they only read them. */
    OSQ_PARAM_SANITIZE = 16,
    /* flag for osq_observe_tokens_fake_quant, OR-ed into `mode`: this call must not use the one-launch PERSISTENT form
     * (a grid of one workgroup per CU whose workgroups wait for each other: it needs the whole device) --
     * three ordinary launches instead, same results.  The per-call counterpart of osq_set_tuning("fused_step", 0). */
    OSQ_PARAM_NO_PERSISTENT = 32
} osq_param_mode;

/* running-statistic rule applied after a batch's (min, max) is known */
typedef enum osq_update_rule {
    OSQ_UPDATE_NONE = 0,     /* only report the batch's (min, max)                      */
    OSQ_UPDATE_RUNNING = 1,  /* observer.py:143-144  min_val = min(min_val, cur) ...    */
    OSQ_UPDATE_AVERAGE = 2   /* observer.py:194-202  (m*cnt + cur) / (cnt+1)            */
} osq_update_rule;

/*
 * Logical [batch, tokens, feat_outer, feat_inner] view of an activation, i.e. the
 * tensor observer.py:72-80 builds with permute+reshape, described by element
 * strides instead of being materialised.  feat_inner is the fastest feature axis.
 *   [B,T,H]            seq_pos=1: outer=1,  inner=H
 *   [B,h,T,d]          seq_pos=2: outer=h,  inner=d
 *   [B,h,d,T]          seq_pos=3: outer=h,  inner=d  (strides of the view)
 *   [B,h,T,S]          seq_pos=2: outer=h,  inner=S
 */
typedef struct osq_token_view {
    int64_t batch, tokens, feat_outer, feat_inner;
    int64_t stride_batch, stride_token, stride_outer, stride_inner;
} osq_token_view;

const char* osq_last_error(void);
int         osq_abi_version(void);
size_t      osq_workspace_bytes(void);

/* Process-wide switches.  The RELEASE library accepts exactly these keys (atomic words, each read once per launch):
 *   summation order (picks WHICH rounding of a whole-tensor sum is returned):
 *     "mse_sum_order" 0 | 8 | 16 | 64   MSEFast losses: 8 / 16 = ATen's one-thread CPU order for 8- / 16-lane SIMD
 *                                       (oracle/aten_sum.py; bit-equal to the reference run on a one-thread host), 64 = double-doubles
 *                                       (order-independent; test mode), 0 = order-free.  The C library starts at 0; the Python
 *                                       host sets 8 when it loads the library (its default tier; OSQ_STRICT=0 keeps 0).  With the
 *                                       order set, per-tensor searches go through osq_msefast_tensor_evals_ordered (one launch per
 *                                       evaluation) or osq_msefast_ordered_multi_* (rounds).
 *     "mse_rows_order" 0 | 8 | 16       the per-channel rows' losses likewise (default 8).
 *     (The other whole-tensor sum of the path, the parameter gradients of the LSQ / LSQ+ backward, takes its order as an
 *     ARGUMENT: osq_lsq_backward_per_tensor_ordered(lanes), osq_lsq_backward_per_channel(sum_lanes).)
 *   path selection (results equal; tests drive every implementation through them):
 *     "fused_step" 0     osq_observe_tokens_fake_quant never uses its one-launch persistent form (per call:
 *                        OSQ_PARAM_NO_PERSISTENT in `mode`; per process also OSQ_FUSED_STEP=0 in the environment);
 *     "mse_resident" 0   per-tensor MSEFast searches of the order-free tier run one launch per evaluation instead of the
 *                        persistent resident form;
 *     "final_fast" 0     the token range finaliser without the two-workgroup kernel; "select_shortcut" 0: that kernel always
 *                        runs its register threshold pass;
 *     "mse_memo" 0       every loss evaluation of a per-tensor MSEFast search streams its tensor, also one whose (scale, zero
 *                        point) pair the search has evaluated before (default 1: answered from the search's memo, see
 *                        osq_msefast_tensor_stats; same results, iterates and nfev either way);
 *   robustness:
 *     "fused_spin_limit" / "mse_spin_limit" n   bound of the cross-workgroup waits of the two persistent launch families
 *                        (0 = the default, ~2 s; n > 0 = n - 1 polls, so 1 makes every wait give up at once: tests force the
 *                        time-out path).
 * Performance A/B knobs -- "fq_unroll", "fq_max_blocks", "fq_nt", "fq_headsplit", "stream_wt", "bwd_blocks", "bwd_order_chunks",
 * "ln_blocks", "obs_blocks", "tok_nt", "fused_gate", "fused_grid", "select_hint", "mse_round_groups" (8), "mse_lean" (1),
 * "mse_grid_all" -- are compile-time constants (the measured winners) in the release library, which answers
 * OSQ_ERR_INVALID_ARGUMENT for them; only the -DOSQ_TUNABLE development build (`make dbg`: libosq_hip_dbg.so) keeps them as
 * variables.  osq_build_flags(): bit 0 = that build, bit 1 = phase timestamps compiled in (-DOSQ_FINAL_TIMING). */
#define OSQ_BUILD_TUNABLE 1
#define OSQ_BUILD_FINAL_TIMING 2
int osq_build_flags(void);
int osq_set_tuning(const char* key, int value);

/* Measurement aid (bench.py).  The events given to osq_time_next_launch ride on the dispatch packet of the
 * NEXT launch of kernel family `which` issued by the calling thread (other launches pass untouched), so
 * osq_timing_elapsed_us(start, stop) is that kernel's own run time on its stream -- the duration
 * rocprofv3 --kernel-trace reports -- not the stream-order interval between two recorded events. */
typedef enum osq_timed_kernel {
    OSQ_TIME_NONE = 0,
    OSQ_TIME_FAKE_QUANT = 1,     /* dense osq_fake_quant_per_tensor                      */
    OSQ_TIME_LSQ_BACKWARD = 2,   /* osq_lsq_backward_per_tensor                          */
    OSQ_TIME_OBSERVE_FLAT = 3,   /* osq_observe_flat (aligned input)                     */
    OSQ_TIME_TOKEN_MINMAX = 4,   /* osq_token_minmax / first launch of osq_observe_tokens */
    OSQ_TIME_TOKEN_SELECT = 5,   /* two-workgroup launch of osq_token_range_finalize      */
    OSQ_TIME_LAYERNORM = 6,      /* osq_residual_layernorm_fake_quant                     */
    OSQ_TIME_FUSED_STEP = 7,     /* one-launch form of osq_observe_tokens_fake_quant      */
    OSQ_TIME_FAKE_QUANT_STRIDED = 8,  /* 16-byte form of osq_fake_quant_per_tensor_strided (head-split views) */
    OSQ_TIME_FAKE_QUANT_CHANNEL = 9,  /* row form of osq_fake_quant_per_channel (weights, ch_axis 0)          */
    OSQ_TIME_OBSERVE_CHANNELS = 10,   /* osq_observe_channels                                                */
    OSQ_TIME_TOKEN_MINMAX_MULTI = 11, /* osq_token_minmax_multi                                              */
    OSQ_TIME_MSEFAST_ROWS = 12,       /* osq_msefast_rows (the per-row search launch)                        */
    OSQ_TIME_OBSERVE_TOKENS = 13      /* one-launch form of osq_observe_tokens                               */
} osq_timed_kernel;
int osq_timing_events_create(void** start, void** stop);
int osq_timing_events_destroy(void* start, void* stop);
int osq_time_next_launch(int which, void* start, void* stop);
int osq_timing_elapsed_us(void* start, void* stop, float* us);

/* ------------------------------------------------------------------ fake-quant forward */

/* util_quant.py:11-15 fake_quantize_per_tensor_affine, as called by
 * FixedFakeQuantize.forward fake_quant.py:123-125 (scale/zero_point stay on device
 * instead of .item()); with mode=LSQ/LSQPLUS also util_quant.py:29-34 / 48-55 forward
 * (LSQFakeQuantize / LSQPlusFakeQuantize.forward, fake_quant.py:159-167 / 199-208).
 * x, y: n contiguous fp32.  x_quant (nullable): the clamped integer tensor, fp32 storage. */
int osq_fake_quant_per_tensor(const float* x, float* y, float* x_quant, int64_t n,
                              float* scale, void* zero_point, int zp_type,
                              int mode, float grad_factor, int quant_min, int quant_max,
                              osq_stream stream);

/* The intermediate-activation site of a transformer block in one pass (model/quant_bert.py:277-280,
 * quant_bart.py:347-348: fc1 / dense -> GELU -> *_act_fn_post_act_fake_quantize): y = fake_quantize(gelu(x))
 * with torch's exact (erf) GELU, bit-identical to F.gelu followed by osq_fake_quant_per_tensor.  x, y: n
 * contiguous fp32, 16-byte aligned (else OSQ_ERR_UNSUPPORTED).  Inference only. */
int osq_gelu_fake_quant_per_tensor(const float* x, float* y, int64_t n,
                                   float* scale, void* zero_point, int zp_type,
                                   int mode, float grad_factor, int quant_min, int quant_max,
                                   osq_stream stream);

/* Same arithmetic for a non-dense tensor of up to 4 dims (e.g. hidden_states[:, 0],
 * quant_bert.py:445): explicit sizes and element strides for input and output. */
int osq_fake_quant_per_tensor_strided(const float* x, float* y, float* x_quant,
                                      const int64_t sizes[4], const int64_t x_strides[4],
                                      const int64_t y_strides[4],
                                      float* scale, void* zero_point, int zp_type,
                                      int mode, float grad_factor, int quant_min, int quant_max,
                                      osq_stream stream);

/* The query / key / value sites of one attention block (model/quant_bert.py:148-155: three Quantizer calls on the
 * head-split views of three [B, T, h*d] projections) as ONE launch.  Site i reads x = [batch, tokens, heads, head_dim]
 * memory (contiguous) and writes y = the dense [batch, heads, tokens, head_dim] tensor the batched matmul wants, with
 * its own (scale, zero_point) -- the same arithmetic, parameter modes (incl. OSQ_PARAM_SANITIZE) and bits as
 * osq_fake_quant_per_tensor_strided on each site.  1..4 sites of ONE geometry; head_dim / 4 a power of two <= 64,
 * 16-byte aligned tensors, else OSQ_ERR_UNSUPPORTED (nothing launched: the caller runs the sites one by one). */
typedef struct osq_headsplit_site {
    const float* x;
    float* y;
    float* scale;
    void* zero_point;
    int32_t zp_type, mode;
    float grad_factor;
    int32_t quant_min, quant_max, pad;
} osq_headsplit_site;
int osq_fake_quant_headsplit_multi(const osq_headsplit_site* sites, int n_sites, int64_t batch, int64_t tokens,
                                   int64_t heads, int64_t head_dim, osq_stream stream);

/* util_quant.py:18-26 fake_quantize_per_channel_affine (and the per-channel learnable
 * forwards :37-45, :58-67).  x is contiguous and viewed as [outer, channels, inner]
 * with ch_axis in the middle; scale/zero_point have `channels` entries. */
int osq_fake_quant_per_channel(const float* x, float* y, float* x_quant,
                               int64_t outer, int64_t channels, int64_t inner,
                               const float* scale, const void* zero_point, int zp_type,
                               int mode, float grad_factor, int quant_min, int quant_max,
                               osq_stream stream);

/* ------------------------------------------------------------------ LSQ / LSQ+ backward */

/* What autograd produces for util_quant.py:48-55 (per-tensor): dx (elementwise, same
 * fp32 operations as autograd), dscale[1], dzero_point[1] (nullable for LSQ).
 * Needed by learn_scale, token_wise_clipping.py:89-107. */
int osq_lsq_backward_per_tensor(const float* x, const float* grad_out, float* grad_x, int64_t n,
                                const float* scale, const void* zero_point, int zp_type,
                                int mode, float grad_factor, int quant_min, int quant_max,
                                float* grad_scale, float* grad_zero_point,
                                void* workspace, osq_stream stream);

/* The same backward in the REFERENCE's summation order: grad_scale / grad_zero_point are the FOUR fp32 reductions autograd
 * forms on the reference's CPU (mul + div backward, add + sub backward; util_quant.py:48-55,70-71), each added in the order
 * of torch.sum on a one-thread host with `lanes` (8 | 16) fp32 SIMD lanes (csrc/aten_order.h), for any n.  The plain entry point above
 * accumulates in float64 and rounds once (order-free; 2e-5 from autograd's fp32 sums) and is ~1.3x faster.  scratch: osq_ordered_sum_scratch_bytes(n, 4) bytes of
 * device memory, contents irrelevant.  x, grad_out, grad_x: contiguous, no alignment requirement. */
size_t osq_ordered_sum_scratch_bytes(int64_t n, int n_sums);
int osq_lsq_backward_per_tensor_ordered(const float* x, const float* grad_out, float* grad_x, int64_t n,
                                        const float* scale, const void* zero_point, int zp_type,
                                        int mode, float grad_factor, int quant_min, int quant_max,
                                        float* grad_scale, float* grad_zero_point, int lanes,
                                        void* scratch, size_t scratch_bytes, void* workspace, osq_stream stream);

/* Per-channel form (util_quant.py:58-67), x viewed as [outer, channels, inner].  sum_lanes 8 | 16: weights
 * (outer == 1, inner <= 3072) take every row's four reductions in torch's one-thread order on that many SIMD lanes (bit-equal to
 * the reference's autograd run); 0 (and every other layout): float64 sums rounded once. */
int osq_lsq_backward_per_channel(const float* x, const float* grad_out, float* grad_x,
                                 int64_t outer, int64_t channels, int64_t inner,
                                 const float* scale, const void* zero_point, int zp_type,
                                 int mode, float grad_factor, int quant_min, int quant_max,
                                 float* grad_scale, float* grad_zero_point,
                                 int sum_lanes, osq_stream stream);

/* fake_quant.py:152-153 / 188-191: what LSQFakeQuantize / LSQPlusFakeQuantize do to their
 * parameters on every forward while the observer is off:
 *   scale.abs_(); scale.clamp_(min=eps); zero_point.clamp_(quant_min, quant_max)  (fp32 zp only)
 * n entries, in place, one launch.  zero_point may be NULL (LSQ keeps an int32 buffer). */
int osq_lsq_sanitize(float* scale, float* zero_point, int64_t n, float eps,
                     int quant_min, int quant_max, osq_stream stream);

/* ------------------------------------------------------------------ observers */

/* observer.py:101-119 ObserverBase.calculate_qparams over n entries.
 * zero_point_out is int32 or fp32 storage according to zp_type. */
int osq_calculate_qparams(const float* min_val, const float* max_val, int64_t n,
                          int quant_min, int quant_max, int symmetric,
                          float* scale_out, void* zero_point_out, int zp_type,
                          osq_stream stream);

/* observer.py:139 torch._aminmax(x) over n contiguous values, followed by the
 * running-statistic rule of the observer (MinMaxObserver :143-144, Avg* :194-202)
 * and, when scale_out != NULL, calculate_qparams (:101-119) -- what
 * QuantizeBase.forward does after observer(...) at fake_quant.py:108-116.
 * cur_minmax (nullable): the batch's own (min, max), 2 floats.  `cnt` is the
 * observer's batch counter BEFORE this call (host int, observer.py:198). */
int osq_observe_flat(const float* x, int64_t n,
                     int update_rule, int64_t cnt, float* min_val, float* max_val,
                     float* cur_minmax,
                     int quant_min, int quant_max, int symmetric,
                     float* scale_out, void* zero_point_out, int zp_type,
                     void* workspace, osq_stream stream);

/* observer.py:141-144 per-channel min/max: _transform_to_ch_axis + _aminmax(y, 1) +
 * running min/max, x contiguous viewed as [outer, channels, inner]; optional
 * qparams per channel.  min_val/max_val hold `channels` entries. */
int osq_observe_channels(const float* x, int64_t outer, int64_t channels, int64_t inner,
                         int update_rule, int64_t cnt, float* min_val, float* max_val,
                         int quant_min, int quant_max, int symmetric,
                         float* scale_out, void* zero_point_out, int zp_type,
                         osq_stream stream);

/* observer.py:64-65 after observer.py:72-84: per-token min and max over the feature
 * axes, for the tokens t < lengths[b] only (lengths == NULL: every token,
 * observer.py:86-98).  token_min/token_max have batch*tokens slots, slot b*T+t;
 * slots of padded tokens are left untouched. */
int osq_token_minmax(const float* x, const osq_token_view* view, const int64_t* lengths,
                     float* token_min, float* token_max, osq_stream stream);

/* osq_token_minmax for MANY tensors in ONE launch (the observer passes of a calibration forward call ~100 sites of 6-50 MB
 * each, token_wise_clipping.py:29-47: launch-bound one by one).  descs / tok_end: DEVICE arrays of n_sites entries; site i
 * is `x` seen through `view`, valid tokens t < lengths[b] (NULL: all), extrema written to token_min / token_max[batch*tokens]
 * (padded slots untouched); vec != 0 promises what osq_token_minmax checks for its 16-byte path (stride_inner == 1,
 * feat_inner % 4 == 0, every other stride % 4 == 0, x 16-byte aligned); tok_end[i] = token slots of sites 0..i. */
typedef struct osq_site_desc {
    const float* x;
    const int64_t* lengths;
    float* token_min;
    float* token_max;
    osq_token_view view;
    int32_t vec;
    int32_t pad;
} osq_site_desc;
int osq_token_minmax_multi(const osq_site_desc* descs, const int64_t* tok_end, int n_sites, int64_t total_tokens,
                           osq_stream stream);

/* Everything after the per-token extrema in AvgPruneMinMaxObserver.forward /
 * AvgMinMaxObserver.forward / MinMaxObserver.forward with a mask:
 *   prune != 0: cac_thres + prune_token (observer.py:50-70) with `percentile`;
 *   prune == 0: plain min/max over the valid tokens (observer.py:193), also used
 *               when 'attention_probs' is in the observer's name (observer.py:62-63);
 * then the update rule (observer.py:194-202 or 143-144) and optional qparams.
 * Up to 32768 token slots, 16-byte aligned arrays, batch*tokens % 4 == 0, tokens >= 4 and a
 * `workspace`: one launch of TWO workgroups, one per side of the statistic (token_max / -token_min),
 * that meet through an 8-byte exchange in the workspace.  Other layouts: one launch, one workgroup.
 * Above 32768 slots, when `workspace` and `list_scratch` (2 * batch * tokens uint32, caller-owned,
 * contents irrelevant) are given: three multi-workgroup launches.  osq_set_wide_min_slots() moves that
 * switch point; osq_set_tuning("final_fast", 0) disables the two-workgroup kernel (tests). */
int osq_token_range_finalize(const float* token_min, const float* token_max,
                             int64_t batch, int64_t tokens, const int64_t* lengths,
                             int prune, double percentile,
                             int update_rule, int64_t cnt, float* min_val, float* max_val,
                             float* cur_minmax,
                             int quant_min, int quant_max, int symmetric,
                             float* scale_out, void* zero_point_out, int zp_type,
                             void* workspace, void* list_scratch, osq_stream stream);

/* osq_token_minmax followed by osq_token_range_finalize on the same stream: a whole masked
 * observation (AvgPruneMinMaxObserver / AvgMinMaxObserver / MinMaxObserver.forward with a mask,
 * observer.py:130-145, 184-203, 214-237) behind one call of the host binding.  token_min/token_max:
 * caller-owned scratch of batch*tokens floats each (their contents are the per-token extrema afterwards). */
int osq_observe_tokens(const float* x, const osq_token_view* view, const int64_t* lengths,
                       float* token_min, float* token_max,
                       int prune, double percentile,
                       int update_rule, int64_t cnt, float* min_val, float* max_val,
                       float* cur_minmax,
                       int quant_min, int quant_max, int symmetric,
                       float* scale_out, void* zero_point_out, int zp_type,
                       void* workspace, void* list_scratch, osq_stream stream);

/* QuantizeBase.forward with observer AND fake-quant enabled (the calibrate-and-quantize state,
 * state.py:22-38; fake_quant.py:107-126 / 178-208) on a masked per-tensor activation, behind one call of the
 * host binding: osq_observe_tokens writing (scale, zero_point), then osq_fake_quant_per_tensor of the same
 * dense x[n] into y with those parameters.  No host sync.  A dense row-major [batch, tokens, features] view
 * with features a multiple of 256 (768, 1024, 3072, 4096), batch <= 1024 and 16-byte aligned buffers runs as ONE
 * persistent launch that keeps the tensor in registers between the reduction and the quantisation (x read from
 * HBM once, fused_step.h; token_min / token_max are scratch whose layout is then private to that launch); anything else, or
 * osq_set_tuning("fused_step", 0), is three launches.  cur_minmax (nullable, 2 floats): also receives THIS batch's
 * (min, max) -- the row a data-parallel calibration exchanges and replays in global batch order (calibration.py; the
 * reference's running mean is sequential, observer.py:194-202) -- while the running statistic moves as usual.
 * osq_fused_step_status reads (and clears) the sticky
 * time-out flags of the one-launch form: 0 = every launch on this workspace completed normally. */
int osq_observe_tokens_fake_quant(const float* x, const osq_token_view* view, const int64_t* lengths,
                                  float* token_min, float* token_max,
                                  int prune, double percentile,
                                  int update_rule, int64_t cnt, float* min_val, float* max_val,
                                  float* cur_minmax,
                                  int quant_min, int quant_max, int symmetric,
                                  float* scale, void* zero_point, int zp_type,
                                  float* y, int64_t n, int mode, float grad_factor,
                                  void* workspace, void* list_scratch, osq_stream stream);

int osq_fused_step_status(void* workspace, int* status_out, osq_stream stream);

/* Sticky time-out flags of BOTH persistent launch families on this workspace -- the one-launch observe + fake-quant step
 * (bit 0: a selector, bit 1: a streaming workgroup waited in vain) and the resident MSEFast searches (bit 0) -- read and
 * cleared after everything enqueued on `stream` has finished (this call synchronises the stream).  Non-zero means some
 * launch since the last call wrote NaN into its outputs because its workgroups were not resident together (another
 * process, or another stream's long kernel, held CUs): the caller must treat every result since the last call as
 * invalid.  reset_on_error != 0 additionally returns both state blocks to their all-zero start, so that the next launch
 * begins clean.  The host binding (outlier_suppression_amd.ops.check_persistent) calls this at its synchronisation
 * points and raises. */
int osq_persistent_status(void* workspace, int* fused_status_out, int* resident_status_out, int reset_on_error,
                          osq_stream stream);

int osq_set_wide_min_slots(int64_t slots);

/* Grid-search form of the same step (token_wise_clipping.py:50-66 calls the observer pass once per
 * candidate percentile although, with fake-quant off, the activations -- hence the per-token
 * extrema -- are identical for every candidate).  The caller keeps the per-token extrema of
 * every (quantizer, batch) pair: problem p = quantizer*n_batches + batch occupies
 * token_min/token_max[p*problem_stride ...], all with the same batch x tokens geometry;
 * lengths is [n_batches, batch] -- or, with lengths_per_quantizer, [n_quantizers, n_batches, batch]: sites of one model
 * may be masked differently (BART's cross-attention keys see the DECODER lengths, quant_bart.py:167,172,472) --;
 * prune_flags[quantizer] == 0 for 'attention_probs'
 * quantizers.  One launch re-thresholds all pairs for `percentile` and writes each pair's
 * (min, max) to cur_table[(batch_index*n_quantizers + quantizer)*2] -- per-batch statistics
 * that are then replayed in batch order with osq_observer_update.  `workspace` (nullable): with it,
 * and the layout rules of osq_token_range_finalize met, every pair gets two workgroups. */
int osq_token_range_finalize_batched(const float* token_min, const float* token_max,
                                     int64_t problem_stride, int n_quantizers, int n_batches,
                                     int64_t batch, int64_t tokens, const int64_t* lengths, int lengths_per_quantizer,
                                     const int32_t* prune_flags, double percentile,
                                     float* cur_table, void* workspace, osq_stream stream);

/* Running statistic for a batch (min, max) that is already known -- the replay step
 * of sharded calibration (gathered per-batch statistics applied in global batch
 * order, observer.py:194-202).  cur_min/cur_max: n entries. */
int osq_observer_update(const float* cur_min, const float* cur_max, int64_t n,
                        int update_rule, int64_t cnt, float* min_val, float* max_val,
                        osq_stream stream);

/* Replay of a whole [n_batches, n_quantizers, 2] table of per-batch (min, max) -- what sharded and cached
 * calibration gather -- in ONE launch: quantizer q folds rows 0..n_batches-1 into its running statistic with
 * rules[q] (OSQ_UPDATE_RUNNING / OSQ_UPDATE_AVERAGE, cnt0 = batches seen before row 0; fresh != 0: start from
 * the untouched (+inf, -inf) state regardless of the buffers), stores it through min_ptrs[q] / max_ptrs[q]
 * (device addresses of 1-element fp32 buffers), and, where scale_ptrs[q] != 0, writes
 * calculate_qparams (observer.py:101-119) through scale_ptrs[q] / zp_ptrs[q] (zp_types[q]).  All arrays live
 * on the device.  Equivalent to n_batches calls of osq_observer_update + n_quantizers of
 * osq_calculate_qparams. */
int osq_replay_statistics(const float* table, int n_batches, int n_quantizers, const int32_t* rules,
                          int64_t cnt0, int fresh, const uint64_t* min_ptrs, const uint64_t* max_ptrs,
                          const int32_t* quant_min, const int32_t* quant_max, const int32_t* symmetric,
                          const uint64_t* scale_ptrs, const uint64_t* zp_ptrs, const int32_t* zp_types,
                          osq_stream stream);

/* Every weight of a model in ONE launch (quantized_module.py:71-72, 97-100 fake-quantise each weight in its own op on
 * every forward).  descs / row_end: DEVICE arrays of n_tensors entries; tensor i is x[rows, inner] (row-major, inner % 4
 * == 0, 16-byte aligned) -> y, quantised per row with scale[row % channels] / zero_point[row % channels] (channels == 1:
 * per-tensor); row_end[i] = rows of tensors 0..i.  mode / grad_factor as in osq_fake_quant_per_channel (no
 * OSQ_PARAM_SANITIZE).  Same arithmetic as the per-tensor calls: bit-identical outputs. */
typedef struct osq_weight_desc {
    const float* x;
    float* y;
    const float* scale;
    const void* zero_point;
    int64_t rows, channels, inner;
    int32_t zp_type, mode;
    float grad_factor;
    int32_t quant_min, quant_max;
    int32_t pad;
} osq_weight_desc;
int osq_fake_quant_weights_multi(const osq_weight_desc* descs, const int64_t* row_end, int n_tensors,
                                 int64_t total_rows, osq_stream stream);

/* ------------------------------------------------------------------ MSEFast (observer.py:412-567) */

/* one_side: 0 = 'no', 1 = 'pos', 2 = 'neg' (observer.py:528-529, decided once by the caller on
 * the first observed tensor); two_d: 1 for the nested range/shift search (observer.py:458-481),
 * 0 for the 1-D search (observer.py:483-494). */

/* golden_section_{1D,2D}_search with ch_axis = 0 (observer.py:496-517): one bounded-Brent
 * search per row of w[rows, cols] -- the reference's Python loop over channels -- in ONE launch.
 * best_min / best_max: rows fp32 entries (the values assigned at observer.py:504,516);
 * nfev (nullable): loss evaluations spent per row. */
int osq_msefast_rows(const float* w, int64_t rows, int64_t cols, int quant_min, int quant_max,
                     int symmetric, int one_side, int two_d,
                     float* best_min, float* best_max, int32_t* nfev, osq_stream stream);

/* Per-tensor search (observer.py:497-499,510-512).  The Brent state machine lives in `state`
 * (osq_msefast_state_bytes() of device memory).  begin: start from the tensor's (min, max)
 * (2 floats on the device, e.g. cur_minmax of osq_observe_flat / osq_token_range_finalize with
 * OSQ_UPDATE_NONE).  evals_*: enqueue n_evals loss evaluations (loss_fx, observer.py:423-432),
 * each one launch that also advances the state machine; launches after convergence are no-ops.
 * done: copy the converged flag to done_out (device int32).  commit: running min/max
 * (observer.py:535-536) or running mean (observer.py:559-567) of the float64 statistics, then
 * calculate_qparams in float64 into scale_out / zero_point_out (nullable).
 * float64_input: the reference casts x to min_val's dtype (observer.py:524,549), and a per-tensor observer's min_val is
 * float64 after its first call (observer.py:481,494) -- so from the second call on its whole search runs on a float64
 * copy of x (float64 fake-quant, float64 mean, np.float64 function values).  1 selects that arithmetic; x stays the
 * fp32 tensor, the kernels widen as they read. */
size_t osq_msefast_state_bytes(void);
int osq_msefast_tensor_begin(void* state, const float* cur_minmax, int quant_min, int quant_max,
                             int symmetric, int one_side, int two_d, int float64_input, osq_stream stream);
int osq_msefast_tensor_evals_flat(void* state, const float* x, int64_t n, int n_evals,
                                  void* workspace, osq_stream stream);
int osq_msefast_tensor_evals_tokens(void* state, const float* x, const osq_token_view* view,
                                    const int64_t* lengths, int n_evals,
                                    void* workspace, osq_stream stream);
/* Loss evaluations in the REFERENCE's summation order ("mse_sum_order" 8 / 16; observer.py:420-432 on a one-thread host):
 * x_flat holds the observed elements in the order remove_padding / flatten lays them out (observer.py:72-84) -- the tensor
 * itself when it is dense and unmasked, otherwise the copy osq_gather_valid_tokens makes (out: batch * tokens * features
 * floats; count_out: device int64 that receives the number of elements kept).  n: elements (an upper bound when n_device,
 * nullable, points at the device-side count).  scratch: osq_ordered_sum_scratch_bytes(n, 1) bytes.  One launch per
 * evaluation; _evals_flat / _evals_tokens refuse to run while the order is set. */
int osq_gather_valid_tokens(const float* x, const osq_token_view* view, const int64_t* lengths, float* out,
                            int64_t* count_out, osq_stream stream);
int osq_msefast_tensor_evals_ordered(void* state, const float* x_flat, int64_t n, const int64_t* n_device, int n_evals,
                                     void* scratch, size_t scratch_bytes, void* workspace, osq_stream stream);
/* The strict evaluations of SEVERAL searches (the MSEFast observers of one forward: independent while fake-quant is off) as
 * ROUNDS: one launch = one loss evaluation of every unfinished search, each in the reference's order as above.  A launch
 * per evaluation of one site costs ~15 us whatever its size; a round pays that once for up to 128 sites.
 *   table: osq_msefast_ordered_multi_bytes(n_sites) bytes of device memory, ZERO before _prepare (site table + the ticket
 *          counters of every site + the workgroup -> site map of a round); _prepare fills table and map (synchronous on
 *          `stream`), copies the device-side element counts into the table (after the gathers enqueued before it) and returns the grid of a round;
 *   per site: state (between osq_msefast_tensor_begin and _commit), the flat tensor (osq_gather_valid_tokens for a masked
 *          site) with its host-side element bound n[i] and nullable device count n_device[i], and scratch of
 *          osq_ordered_sum_scratch_bytes(n[i], 1) bytes;
 *   _evals: n_evals rounds, then done_out[0] = 1 if every search has converged (nullable). */
size_t osq_msefast_ordered_multi_bytes(int n_sites);
int osq_msefast_ordered_multi_prepare(void* table, size_t table_bytes, void* const* states, const float* const* x_flat,
                                      const int64_t* n, const int64_t* const* n_device, void* const* scratch,
                                      const size_t* scratch_bytes, int n_sites, int* total_blocks_out, osq_stream stream);
int osq_msefast_ordered_multi_evals(const void* table, int n_sites, int total_blocks, int n_evals, int32_t* done_out,
                                    osq_stream stream);
/* The whole per-tensor search in ONE persistent launch (between _begin and _commit; replaces the _evals_* / _done loop):
 * the valid part of the tensor is loaded once into the registers of a one-workgroup-per-CU grid, every loss
 * evaluation is one exchange of per-workgroup partial sums through the workspace.  view == NULL: x is flat, n
 * elements; otherwise a token view with valid lengths (n ignored).  OSQ_ERR_UNSUPPORTED (nothing launched): more
 * than 32 float4 per lane of the grid (512-thread workgroups: 67 MB on 256 CUs), batch > 512, a misaligned flat tensor,
 * or osq_set_tuning("mse_resident", 0) -- the caller then runs the _evals_* loop.  A workgroup that waits in vain for
 * another one's partial sum (the grid was not resident together: another process or stream held CUs) poisons the search
 * with NaN and raises the sticky flag osq_persistent_status reports. */
int osq_msefast_tensor_search(void* state, const float* x, int64_t n, const osq_token_view* view,
                              const int64_t* lengths, void* workspace, osq_stream stream);
/* Several searches in ONE persistent launch (the observers of a forward are independent while fake-quant is off):
 * osq_msefast_resident_slots(elems) = float4 slots per lane of the resident grid a search over `elems` elements takes
 * (1..max_slots; 0 = cannot be resident); a group fits when it has at most max_sites searches whose slots add up to at
 * most max_slots (osq_msefast_resident_limits: 32 and 16).
 * Every round of the launch evaluates the pending candidate of every unfinished search; the exchange of the partial
 * sums and the serial Brent steps of the searches overlap.  Arrays are HOST arrays of n_sites entries; views[i].batch
 * == 0 marks a flat tensor of ns[i] elements.  Each search sits between its own _begin and _commit.
 * OSQ_ERR_UNSUPPORTED: the group does not fit, nothing was launched. */
int osq_msefast_resident_slots(int64_t elems);
int osq_msefast_resident_limits(int* max_slots, int* max_sites);
int osq_msefast_tensor_search_multi(void* const* states, const float* const* xs, const int64_t* ns,
                                    const osq_token_view* views, const int64_t* const* lengths, int n_sites,
                                    void* workspace, osq_stream stream);
int osq_msefast_tensor_done(const void* state, int32_t* done_out, osq_stream stream);
/* The LOSS MEMO of the per-tensor searches (ABI 7).  loss_fx (observer.py:423-432) is a pure function of the tensor and of the
 * pair it hands to the fake-quant -- scale.item(), int(zero_point.item()) -- and scipy's bounded search asks for the same pair
 * many times (the inner search of observer.py:434-446 moves the shift of a fixed range: the scale stays, the integer zero point
 * changes once per quantisation step; the final inner search of observer.py:469-475 repeats an earlier one entirely).  A search
 * keeps the pairs it has streamed (up to 512) in its state; the step after an evaluation advances the state machine for as long
 * as the next candidate's pair is known.  Results, iterates and nfev are those of the search without it, bit for bit; on
 * two-sided data 3/4 of the passes over the tensor go.  osq_set_tuning("mse_memo", 0) switches it off (tests, A/B).
 * The _evals_flat / _evals_tokens / _evals_ordered / _ordered_multi_evals entry points use it; the persistent searches
 * (osq_msefast_tensor_search*: the tensor is in registers, an evaluation costs microseconds) do not.
 * stats_out: 4 device int32 -- nfev so far, pairs kept, evaluations answered from the memo, converged flag. */
int osq_msefast_tensor_stats(const void* state, int32_t* stats_out, osq_stream stream);
/* ObserverBase.calculate_qparams (observer.py:101-119) on float64 statistics -- what a per-tensor MSEFast observer holds:
 * torch computes in the statistics' (promoted) dtype; scale / zero_point are rounded to their fp32 / int32 storage once. */
int osq_calculate_qparams_f64(const double* min_val, const double* max_val, int64_t n, int quant_min, int quant_max,
                              int symmetric, float* scale_out, void* zero_point_out, int zp_type, osq_stream stream);

/* ref_float64 (nullable): int32[2] on the device, zero before the observer's first commit -- whether the REFERENCE's
 * min_val / max_val hold float64 by now.  Its per-tensor results are float64 except the float32 zeros_like of a one-sided
 * search (observer.py:491-492) and the float32 extremum that Python's max / min hand back when the nested search's range
 * reaches beyond the data (observer.py:479-480); while a statistic is float32 its running mean is fp32 arithmetic, while
 * both are, calculate_qparams is, and observer.py:524 / 549 cast the NEXT batch to min_val's dtype: the caller reads
 * ref_float64[0] for osq_msefast_tensor_begin's float64_input.  NULL: float64 throughout. */
int osq_msefast_tensor_commit(const void* state, int update_rule, int64_t cnt,
                              double* min_val, double* max_val,
                              int quant_min, int quant_max, int symmetric,
                              float* scale_out, void* zero_point_out, int zp_type,
                              int32_t* nfev, int32_t* ref_float64, osq_stream stream);

/* ------------------------------------------------------------------ remaining observers of ObserverDict */

/* LSQPlusObserver.forward (observer.py:159-173): min/max = mean -+ 3*std (unbiased) of x viewed as
 * [outer, channels, inner]; channels == 1 is the per-tensor form (needs workspace).  Not accumulated
 * across calls, as in the reference.  Optional qparams. */
int osq_observe_moments(const float* x, int64_t outer, int64_t channels, int64_t inner,
                        float* min_val, float* max_val, int quant_min, int quant_max, int symmetric,
                        float* scale_out, void* zero_point_out, int zp_type,
                        void* workspace, osq_stream stream);

/* AvgQuantileObserver.forward (observer.py:253-282) after the tensor's own (min, max) is known
 * (cur_minmax, 2 device floats): torch.histc(|x|, 2048, 0, max(-min, max)) with torch's binning
 * (linear estimate + local search against the linspace edges), clip at the bin where the
 * cumulative count reaches threshold * numel, running mean, optional qparams.  x is either flat
 * (view == NULL, n elements) or a token view with valid lengths.  hist_scratch: 2048 uint32,
 * zero on entry, left zeroed. */
int osq_observe_quantile(const float* x, int64_t n, const osq_token_view* view, const int64_t* lengths,
                         const float* cur_minmax, double threshold, uint32_t* hist_scratch,
                         int update_rule, int64_t cnt, float* min_val, float* max_val,
                         int quant_min, int quant_max, int symmetric,
                         float* scale_out, void* zero_point_out, int zp_type, osq_stream stream);

/* MSEObserver / AvgMSEObserver (observer.py:285-409): brute-force grid of 100 clipping ranges
 * (1-D, symmetric or one-sided data) or 100 ranges x (quant_max-quant_min+1) zero-points (2-D);
 * 32 candidates are evaluated per pass over the data.  loss_scratch: scratch_bytes bytes of device memory; with at least
 * osq_mse_grid_scratch_bytes() (13 MB for the 6-bit asymmetric grid: the losses + 256 workgroups' partial sums of every
 * candidate) the whole grid is ONE launch of 1024-thread workgroups + a reduction; with less (>= osq_mse_grid_candidates()
 * floats) it is one launch per 32 candidates.  The first osq_mse_grid_candidates() floats hold the losses afterwards. */
int osq_mse_grid_candidates(int quant_min, int quant_max, int two_d);
size_t osq_mse_grid_scratch_bytes(int quant_min, int quant_max, int two_d);
int osq_mse_grid_tensor(const float* x, int64_t n, const osq_token_view* view, const int64_t* lengths,
                        const float* cur_minmax, int quant_min, int quant_max, int symmetric,
                        int one_side, int two_d, float* loss_scratch, size_t scratch_bytes,
                        int update_rule, int64_t cnt, float* min_val, float* max_val,
                        float* scale_out, void* zero_point_out, int zp_type,
                        void* workspace, osq_stream stream);
/* Test aid: the all-candidates launch replaces x / scale by a reciprocal sequence that is bit-equal to the IEEE division under
 * two guards (csrc/observers_extra.hip, div_by_reciprocal); this counts the admitted pairs (x[i], s[i]) on which the two differ. */
int osq_selftest_division(const float* x, const float* s, int64_t n, int32_t* mismatches, osq_stream stream);
/* per-channel form (observer.py:297-301,316-323): one wave per row of w[rows, cols]. */
int osq_mse_grid_rows(const float* w, int64_t rows, int64_t cols, int quant_min, int quant_max,
                      int symmetric, int one_side, int two_d,
                      float* best_min, float* best_max, osq_stream stream);

/* ------------------------------------------------------------------ gamma migration */

/* gamma_migration.py:70-71  w.weight.data *= gamma  (gamma broadcast over columns). */
int osq_gamma_fold(float* weight, const float* gamma, int64_t rows, int64_t cols,
                   osq_stream stream);

/* util_layernorm.py:27  bias <- beta / gamma. */
int osq_gamma_split_bias(const float* beta, const float* gamma, float* bias_out, int64_t n,
                         osq_stream stream);

/* util_layernorm.py:49-52 GammaResidual.forward: out = input * gamma + hidden
 * (gamma == NULL: input + hidden).  rows x cols contiguous. */
int osq_gamma_residual(const float* input, const float* hidden, const float* gamma,
                       float* out, int64_t rows, int64_t cols, osq_stream stream);

/* ------------------------------------------------------------------ residual + LayerNorm + fake-quant */

/* One pass for what a LayerNorm site of the quantized models runs as four eager steps
 * (model/quant_bert.py:211-216 / 298-303 with model/util_layernorm.py:14-18, 32-37, 49-52):
 *     r = x * gamma + hidden        GammaResidual.forward      (hidden == NULL: r = x; gamma == NULL: x + hidden)
 *     n = layer_norm(r, eps)        over the last axis, cols
 *     n = n * weight + bias         QuantizedLayerNorm: the LayerNorm's affine pair;
 *                                   QuantizedSplitLayerNorm: weight == NULL, bias = beta/gamma
 *     y = fake_quantize(n)          scale == NULL: y = n (observer passes, qoutput == False)
 * rows x cols contiguous fp32, cols % 4 == 0, cols <= 4096, 16-byte aligned operands (else
 * OSQ_ERR_UNSUPPORTED and the caller keeps the eager sequence).  mode / grad_factor / zp_type as in
 * osq_fake_quant_per_tensor, including OSQ_PARAM_SANITIZE.  8 B per element (12 with a residual)
 * instead of 24-32.  Inference only: autograd passes use the eager sequence. */
int osq_residual_layernorm_fake_quant(const float* x, const float* hidden, const float* gamma,
                                      const float* weight, const float* bias, double eps,
                                      float* y, int64_t rows, int64_t cols,
                                      float* scale, void* zero_point, int zp_type,
                                      int mode, float grad_factor, int quant_min, int quant_max,
                                      osq_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OSQ_HIP_H */
