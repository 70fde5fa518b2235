/* simclr_hip.h -- C ABI of libsimclr_hip.so: the MI355X (gfx950) SimCLR pretraining hot path.
 *
 * The reference (google-research/simclr, tf2/) has no native/FFI boundary: the hot path sits
 * behind Python callables that hand everything to TensorFlow ops.  This header is the boundary a
 * maintainer binds instead (ctypes stub in INTEGRATION.md); every entry point names the reference
 * lines whose arithmetic it replaces.  Paths are relative to /root/reference/.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / framework types.
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise; the library
 *     never allocates, frees or synchronises; work is enqueued on `stream` (a hipStream_t).
 *   - returns 0 on success; nonzero = error (1 bad argument, 2 launch failure) with a message in
 *     simclr_last_error() (thread-local).  Re-entrant across streams.
 *   - `dtype`: SIMCLR_DT_F32 (0) or SIMCLR_DT_BF16 (1) = storage type of activations / compute
 *     weights ("T" below).  Accumulation, statistics, losses and optimizer state are fp32/fp64.
 *   - layouts: activations NHWC, master conv weights HWIO fp32, dense weights [in,out] fp32
 *     (tf2/resnet.py:196-203, tf2/model.py:143-146).
 *   - `dtype` may carry OPTION FIELDS above the low byte (fp32 storage only; entry points that
 *     accept them say so):
 *       SIMCLR_FMT_TERMS(t)  bits 12..19: the matrix arithmetic of THIS call (t = 0 exact fp32 MFMA,
 *                            3 | 6 bf16-piece terms, 13 = three fp16-piece terms, forward only) -- the
 *                            process-wide default of simclr_set_f32_matmul is then not consulted, so
 *                            calls on different streams / threads can run different modes;
 *       SIMCLR_FMT_PS_IN     the gradient operand (dy) of simclr_conv2d_dgrad / _dgrad_bn / _wgrad is
 *                            in the PRE-SPLIT BLOCK FORMAT: same 4 bytes per element, but every 128-byte
 *                            block of 32 channels holds eight 16-byte chunks of bf16 pieces -- chunk g
 *                            (g < 4) = hi = bf16(x) of channels {4g..4g+3, 16+4g..16+4g+3}, chunk 4 + g =
 *                            lo = bf16(x - hi) of the same channels -- so that the three-term GEMMs read
 *                            their operand pieces without splitting anything in the k-loop;
 *       SIMCLR_FMT_PS_OUT    simclr_bn_bwd_apply / simclr_bn_bwd_apply_pool write dx in that format (channels % 32 == 0);
 *       SIMCLR_FMT_PS_W      the WEIGHT operand of simclr_conv2d_fwd / _fwd_pivoted / _fwd_bn_apply (terms 13) or of
 *                            simclr_conv2d_dgrad / _dgrad_bn (terms 3) is the pre-split copy simclr_presplit_weights_multi
 *                            made of it (once per optimizer step instead of once per launch inside the library).
 */
#ifndef SIMCLR_HIP_H_
#define SIMCLR_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMCLR_DT_F32 0
#define SIMCLR_DT_BF16 1
#define SIMCLR_FMT_PS_IN 0x100
#define SIMCLR_FMT_PS_OUT 0x200
#define SIMCLR_FMT_TERMS(t) (((t) + 1) << 12)
#define SIMCLR_FMT_PS_W 0x100000

typedef struct ihipStream_t* simclr_stream_t; /* == hipStream_t */

/* ---- runtime ---------------------------------------------------------------------------------- */
const char* simclr_last_error(void);
int simclr_abi_version(void);
/* lane-layout probes used by the GPU tests (0 mfma bf16 16x16x32, 1 mfma f32 16x16x4, 2 ds_read_tr16, 3 mfma f16 16x16x32 incl. subnormal inputs) */
int simclr_probe(int which, const void* a, const void* b, void* out, simclr_stream_t stream);

/* ---- NT-Xent contrastive loss: tf2/objective.py:35-89 (add_contrastive_loss) ------------------ */
/* tf.math.l2_normalize, objective.py:53-54.  z = x*rsqrt(max(sum x^2,1e-12)); inv[row] kept for bwd. */
int simclr_l2norm_fwd(const float* x, float* z, float* inv, int rows, int D, simclr_stream_t stream);
int simclr_l2norm_bwd(const float* z, const float* inv, const float* dz, float* dx, int rows, int D,
                      simclr_stream_t stream);
/* Matrix arithmetic of the sweeps (round 6, opt-in): the D argument of simclr_ntxent_fwd / _bwd may carry SIMCLR_FMT_TERMS(13) --
 * every fp32 product of S = Q K^T and of dF += T^T dS then runs as three fp16-piece MFMA terms (~2^-22 per product, i.e.
 * ~2^-22 / T on the logits) instead of the fp32-input MFMA (1/16 of the 16-bit rate).  Only for l2-NORMALISED rows (|z| <= 1 lies
 * in fp16's range; tf2/objective.py:53-54 hidden_norm=True); both calls of one loss must use the same setting. */
size_t simclr_ntxent_workspace_bytes(int n, int N, int D);
/* objective.py:55-87 + metrics.py:28-31.  z_local [2n,D] = [hidden1;hidden2] of this replica,
 * z_all [2N,D] = [hidden1_large;hidden2_large] (objective.py:60-61), N = R*n, rank = replica id
 * (objective.py:64-67).  out[0] = loss, out[1] = contrast_acc.  row_stats [2n,2] feeds the bwd. */
int simclr_ntxent_fwd(const float* z_local, const float* z_all, int n, int N, int D, int rank,
                      float temperature, float* out, float* row_stats, void* workspace,
                      simclr_stream_t stream);
/* What tape.gradient (tf2/run.py:621) derives for objective.py:76-87.  dz_local [2n,D]: gradient
 * through the local (query) rows; dz_all [2N,D]: gradient through the gathered rows, to be
 * reduce-scattered (SUM) across replicas = transpose of objective.py:114-122.  grad_scale = the
 * upstream factor (1/R, tf2/run.py:617).  out[2] = contrast_entropy (metrics.py:33-35). */
int simclr_ntxent_bwd(const float* z_local, const float* z_all, int n, int N, int D, int rank,
                      float temperature, const float* row_stats, float grad_scale, float* dz_local,
                      float* dz_all, float* out, void* workspace, simclr_stream_t stream);
/* dense logits_ab [n,N] (objective.py:80,89) for API parity only; not on the hot path. */
int simclr_ntxent_logits_ab(const float* z_local, const float* z_all, int n, int N, int D,
                            float temperature, float* logits_ab, simclr_stream_t stream);

/* ---- LARS: tf2/lars_optimizer.py:83-137 (_resource_apply_dense), all tensors in 2 launches ---- */
/* table: device int64[5*T] = {w ptrs | g ptrs | v ptrs | numel | flags(bit0 use_weight_decay :139-148,
 * bit1 do_layer_adaptation :150-157)}; chunks: device int64[2*num_chunks] = (tensor id, element
 * offset), one per simclr_lars_chunk_elems() elements, a tensor's chunks consecutive and in offset order; norms: device
 * double[2*num_chunks] scratch (per-chunk partial norms, summed in a fixed order: the update is run-to-run deterministic).
 * lr_dev (nullable) overrides lr with a device-resident value (graph replay). */
int simclr_lars_chunk_elems(void);
int simclr_lars_multi_tensor(const long long* table, int num_tensors, const long long* chunks,
                             int num_chunks, const float* lr_dev, float lr, float momentum,
                             float weight_decay, float eeta, int classic_momentum, int use_nesterov,
                             double* norms, simclr_stream_t stream);
/* The other two branches of build_optimizer (tf2/model.py:31-34), same descriptor / chunk tables (row 2 = the slot; flags
 * bit 0 = add l2 * w to the gradient: the derivative of add_weight_decay's loss term, tf2/model.py:62-69).
 * simclr_sgd_multi_tensor: tf.keras.optimizers.SGD(lr, momentum, nesterov): accum = momentum * accum - lr * g;
 * w += nesterov ? momentum * accum - lr * g : accum.  simclr_adam_multi_tensor: tf.keras.optimizers.Adam (the slot holds
 * 2 * numel floats: m, then v); step = 1-based update count of the bias correction. */
int simclr_sgd_multi_tensor(const long long* table, int num_tensors, const long long* chunks, int num_chunks,
                            const float* lr_dev, float lr, float momentum, int use_nesterov, float l2, simclr_stream_t stream);
int simclr_adam_multi_tensor(const long long* table, int num_tensors, const long long* chunks, int num_chunks,
                             const float* lr_dev, float lr, float beta1, float beta2, float epsilon, long long step, float l2,
                             simclr_stream_t stream);

/* ---- convolution / dense: tf2/resnet.py:183-208 (Conv2dFixedPadding), tf2/model.py:143-154 ----- */
/* Matrix arithmetic of the SIMCLR_DT_F32 convolution / dense launches (the reference's tf.nn.conv2d / tf.matmul on float32,
 * resnet.py:196-208, model.py:148-153): number of bf16 terms per fp32 product, for the forward GEMM and for the two backward
 * GEMMs (dgrad, wgrad).  0 (default) = exact fp32 MFMA; 3 = x_hi*w_hi + x_hi*w_lo + x_lo*w_hi (~2^-17 per product);
 * 6 = every term of weight >= 2^-18 of a three-way split (fp32 level).  Storage, accumulation, statistics and every
 * elementwise kernel stay fp32; bf16 launches are unaffected.  fwd_terms = 13 (round 6): three FP16-piece terms
 * (11-bit pieces: ~2^-22 per product, the accuracy of six bf16 terms at half the MFMA work; operands must lie within fp16's
 * range -- BatchNorm outputs, images, weights do; an out-of-range operand gives inf / NaN, never a silently wrong value; the
 * stem keeps six bf16 terms).  This call sets the process-wide DEFAULT only: a convolution / dense entry point whose `dtype`
 * carries SIMCLR_FMT_TERMS(t) runs with t whatever the default is (the re-entrant form; simclr_amd/ops.py always sends it).
 * The setting is a LOWER bound on accuracy: the non-persistent fallback kernel (debug switch SIMCLR_NO_PERSISTENT, or more than
 * 64 N-tiles) always runs the exact fp32 MFMA whatever is selected here.
 * simclr_get_f32_matmul(0 | 1) returns the forward | backward setting. */
int simclr_set_f32_matmul(int fwd_terms, int bwd_terms);
int simclr_get_f32_matmul(int which);
/* Scheduling of simclr_conv2d_fwd / _dgrad / _dgrad_bn / _fwd_bn_apply (no reference counterpart: TensorFlow's runtime
 * schedules its own kernels): the persistent grid runs floor(tiles / workgroups) whole output tiles per workgroup and
 * shares every left-over tile between 2..8 workgroups along the reduction ("split tail"); the partial fp32 accumulators
 * meet in a scratch buffer the LIBRARY owns -- the one exception to "never allocates": <= 64 MB per stream, hipMalloc'ed
 * on the first launch that needs it (outside any graph capture), kept for the life of the process.  Results are
 * deterministic (fixed summation order).  Environment SIMCLR_IGEMM_SPLIT=0 disables it.  The hook below returns the
 * number of parts the most recent launch used (0 = whole tiles only); tests use it to prove the path ran. */
int simclr_conv2d_last_split_parts(void);
/* The owner of a shared tile waits for its partners with a BOUNDED spin (a lost partner must not hang the device).  A partner
 * that never arrives is counted in a sticky device word next to the flags; this call copies the counters back (synchronous --
 * call it where the host synchronises anyway: simclr_amd/run.py does at every logging interval) and returns their sum in
 * *host_total: 0 on a healthy device.  The flags are reset by the consumer inside the launch and carry no host-side sequence
 * number, so a launch captured into a hipGraph replays correctly (the scratch must exist before the capture starts). */
int simclr_conv2d_split_tail_timeouts(unsigned* host_total);
/* Which kernel instantiation the most recent forward / dgrad launch of this process selected: the template-argument list as
 * spelled at the launch site plus element size, GEMM role, grid, block, LDS bytes, tile counts and split-tail parts.  With the
 * environment variable SIMCLR_DRY_RUN=1 the convolution entry points take every launch decision and record it here WITHOUT
 * touching the device (pointers are not dereferenced, nothing is allocated or launched): tools/instantiation_table.py dumps the
 * (layer class -> instantiation) table of the benchmark models on a machine without a GPU. */
const char* simclr_conv2d_last_instantiation(void);
/* Three-term data gradient of the SIMCLR_DT_F32 launches (simclr_set_f32_matmul(*, 3)): the weight operand is rewritten once per
 * launch into (hi, lo) bf16 planes in a library-owned per-stream buffer (the second exception to "never allocates": the
 * largest weight matrix, >= 16 MB, hipMalloc'ed on first use), so the k-loop does no splitting work for it; bitwise the
 * same result as the in-register split.  Environment SIMCLR_F32_PRESPLIT=0 disables it.  The hook returns 1 if the most
 * recent fp32 data-gradient launch took that path. */
int simclr_conv2d_last_presplit(void);
/* master HWIO fp32 -> compute copies.  mode 0: [Cout][KH*KW*Cin] (fwd), 1: [Cin][KH*KW*Cout]
 * (dgrad), 2: stem [Cout][KHP][KWP][4] zero padded.  CinP/CoutP (0 = none): zero-padded channel dims. */
int simclr_prep_weights(const float* w_hwio, void* dst, int KH, int KW, int Cin, int Cout, int mode,
                        int KHP, int KWP, int CinP, int CoutP, int dtype, simclr_stream_t stream);
/* modes 0 (dst_t) and 1 (dst_d) of the same weight in one launch. */
int simclr_prep_weights_pair(const float* w_hwio, void* dst_t, void* dst_d, int KH, int KW, int Cin, int Cout,
                             int CinP, int CoutP, int dtype, simclr_stream_t stream);
/* The (dst_t, dst_d) pairs of MANY convolutions in one launch: the refresh of every compute copy after an optimizer
 * step (tf2/resnet.py:183-208 has 52 of these layers in ResNet-50).  table: device int64 [T][8] = {w_hwio, dst_t, dst_d,
 * KH*KW, Cin, Cout, CinP, CoutP}; chunks: device int64 [nchunks][2] = {tensor index, tile}: one entry per E x E tile
 * (E = simclr_prep_chunk_elems()) of every tap's padded (CinP, CoutP) plane, tile = (tap * nci + ci_tile) * nco + co_tile,
 * nci = ceil(CinP / E), nco = ceil(CoutP / E). */
int simclr_prep_chunk_elems(void);
int simclr_prep_weights_pair_multi(const long long* table, const long long* chunks, int nchunks, int dtype,
                                   simclr_stream_t stream);
/* Partial-statistics slots.  Every `stats` / `partial` argument below is float [nslot][2][C], zeroed by the caller.
 * With nslot >= the number returned here each producing workgroup stores into its OWN slot (no float atomics) and the
 * slot reduction of simclr_bn_finalize / simclr_bn_bwd_finalize / simclr_bn_reduce_slots adds the slots in a fixed
 * order: BatchNorm statistics are then bit-identical from run to run, like the reference's SyncBatchNormalization
 * (tf2/resnet.py:54-60).  With fewer slots the producers fall back to float atomics into slot (workgroup % nslot). */
int simclr_conv2d_stats_slots(long long M, int C);      /* simclr_conv2d_fwd, simclr_conv2d_dgrad_bn: M output rows x C channels */
int simclr_stem_stats_slots(long long M);               /* simclr_stem_conv_fwd: M = V*OH*OW */
int simclr_bn_bwd_reduce_slots(long long rows, int C, int dtype);

/* y[V,OH,OW,Cout] = conv(x[V,IH,IW,Cin], w), explicit symmetric padding `pad` (resnet.py:167-180).
 * stats (nullable) float[nslot][2][Cout], zeroed by caller: per-channel partial (sum, sum sq) of y
 * for BatchNorm (resnet.py:50-78) -- of the fp32 accumulators (128-wide tiles, eight-phase 256-wide tile); the 256 x 256
 * instantiation of the persistent kernel (SIMCLR_IGEMM_TILE=256 / K-extended dgrad classes) takes them over the bf16-ROUNDED
 * outputs, the tensor the reference's moments see: the two differ by the output rounding (<= 2^-9 relative per element,
 * ~1e-5 sigma on the mean at 10^5 rows).  Cin % 64 == 0 (bf16) / 32 (f32); Cout % 4 == 0.
 * y == NULL (bf16, stats required): statistics-only pass -- the convolution is computed, nothing is stored. */
int simclr_conv2d_fwd(const void* x, const void* w_t, void* y, float* stats, int nslot, int V, int IH,
                      int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad,
                      int dtype, simclr_stream_t stream);
/* simclr_conv2d_fwd for SIMCLR_DT_F32 with PIVOTED BatchNorm statistics (resnet.py:50-78: the moments of the convolution
 * output).  pivot float[Cout] (out): the convolution output at one interior pixel (image 0, OH/2, OW/2); the slots receive
 * sum(y - pivot[n]) and sum((y - pivot[n])^2) over the valid rows, and simclr_bn_reduce_slots_pivoted(count = V*OH*OW) turns
 * them into the raw fp64 moments simclr_bn_finalize(sums) takes.  Raw fp32 moments lose (mean / sigma)^2 * 2^-24 of the
 * variance; about a pivot the sums stay at the scale of the spread.  y, stats and pivot are required. */
int simclr_conv2d_fwd_pivoted(const void* x, const void* w_t, void* y, float* stats, int nslot, float* pivot, int V,
                              int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad,
                              int dtype, simclr_stream_t stream);
/* The same convolution with its consumer's BatchNorm apply in the epilogue (bf16 only; resnet.py:470-487, conv3 -> bn3 ->
 * + shortcut -> relu):  y = act(bf16(conv(x)) * scale + shift + r), r = res or res * rscale + rshift (projection shortcut
 * whose BatchNorm is applied here too, resnet.py:411-421);  relu_bits[i] bit e = (y[8 i + e] > 0).
 * Bitwise equal to simclr_conv2d_fwd followed by simclr_bn_apply; the convolution output never reaches memory.
 * scale / shift [Cout]: from simclr_bn_finalize over the statistics of a first simclr_conv2d_fwd(y = NULL) pass.
 * res (nullable) [V,OH,OW,Cout]; rscale / rshift (nullable, both or none) [Cout]; relu: 0 | 1; relu_bits (nullable)
 * uint8 [V*OH*OW*Cout/8]. */
int simclr_conv2d_fwd_bn_apply(const void* x, const void* w_t, void* y, const float* scale, const float* shift,
                               const void* res, const float* rscale, const float* rshift, int relu,
                               unsigned char* relu_bits, int V, int IH, int IW, int Cin,
                               int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dtype,
                               simclr_stream_t stream);
/* dx[V,IH,IW,Cin] (+)= conv_transpose(dy[V,OH,OW,Cout], w); autodiff of the above (run.py:621). */
int simclr_conv2d_dgrad(const void* dy, const void* w_d, void* dx, int accumulate, int V, int IH, int IW,
                        int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dtype,
                        simclr_stream_t stream);
/* dgrad whose output is the gradient wrt a BatchNormRelu output (tf2/resnet.py:74-78): the ReLU mask and
 * the BatchNorm-backward reductions sum(dm), sum(dm*x^) are fused into the epilogue; dx receives
 * dm = dx*mask.  stats float[nslot][2][Cin] zeroed by the caller.  mask_mode 1: bn_mask>0 (tensor after the
 * ReLU, e.g. the block output of resnet.py:487); 2: bn_x*scale+shift>0; 3: bn_mask = the bit tensor written by
 * simclr_bn_apply(relu_bits) (16x less traffic than mode 1); 4: as 3 but ONLY sum(dm) is produced and bn_x / bn_mean /
 * bn_rstd are not read (may be NULL) -- sum(dm*x^) then comes from simclr_bn_fold_s2.  stride 1 only. */
int simclr_conv2d_dgrad_bn(const void* dy, const void* w_d, void* dx, int accumulate, const void* bn_x,
                           const void* bn_mask, const float* bn_scale, const float* bn_shift,
                           const float* bn_mean, const float* bn_rstd, int mask_mode, float* stats, int nslot,
                           int V, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW,
                           int stride, int pad, int dtype, simclr_stream_t stream);
size_t simclr_conv2d_wgrad_workspace_bytes(int V, int OH, int OW, int Cin, int Cout, int KH, int KW,
                                           int dtype);
/* dw[KH*KW*Cin][Cout] fp32 (HWIO) (+)= x^T * dy.  `pixpitch` = elements between neighbouring pixels
 * of x (== Cin except for the packed stem input). */
int simclr_conv2d_wgrad(const void* x, const void* dy, float* dw, int accumulate, void* workspace, int V,
                        int IH, int IW, int Cin, int pixpitch, int OH, int OW, int Cout, int KH, int KW,
                        int stride, int pad, int dtype, simclr_stream_t stream);
/* stem conv (Cin=3; resnet.py:593-599 7x7 s2, :551-556 CIFAR 3x3 s1) on the packed input */
int simclr_stem_conv_fwd(const void* xp, const void* w_s, void* y, float* stats, int nslot, int V, int HP,
                         int WP, int OH, int OW, int Cout, int KHP, int KWP, int stride, int dtype,
                         simclr_stream_t stream);
int simclr_unpack_stem_dw(const float* src, float* dst, int KH, int KW, int Cin, int Cout, int KWP,
                          int accumulate, simclr_stream_t stream);
/* Stem weight gradient of the three-term parity mode (fp32 storage, three bf16-piece backward terms) from PRE-SPLIT operands
   (tf2/resnet.py:593-599 under tape.gradient): xq = simclr_presplit_packed(xp) -- every packed pixel [4] fp32 becomes four bf16 hi
   pieces followed by four bf16 lo pieces, same 16 bytes --, dy_ps [V*OH*OW][64] in the pre-split block format
   (simclr_bn_bwd_apply with SIMCLR_FMT_PS_OUT).  dw_kn: fp32 [KH*KWP*4][64], the layout simclr_unpack_stem_dw reads; workspace:
   simclr_stem_wgrad_ps_workspace_bytes.  Only the 7x7 / stride-2 stem with 64 output channels (simclr_stem_wgrad_ps_supported);
   every other stem keeps simclr_conv2d_wgrad on the packed input. */
int simclr_presplit_packed(const void* xp, void* xq, long long npix, simclr_stream_t stream);
/* Pre-split copies of n compute-weight matrices in ONE launch (what the three-term forward / data-gradient launches otherwise make of
   their weight operand per call): table [n][4] int64 on the device = (source fp32 [rows][K], K % 32 == 0; destination, same size;
   128-byte k-blocks = rows * K / 32; pieces: 0 = bf16 -- a data-gradient call's w_d --, 1 = fp16 of 2^8 * w -- a forward call's w_t);
   max_blocks = the largest block count in the table.  Pass the destination as the weight argument together with SIMCLR_FMT_PS_W. */
int simclr_presplit_weights_multi(const long long* table, int n, long long max_blocks, simclr_stream_t stream);
int simclr_stem_wgrad_ps_supported(int KH, int KWP, int stride, int Cout);
size_t simclr_stem_wgrad_ps_workspace_bytes(int V, int OH, int OW, int KH);
int simclr_stem_wgrad_ps(const void* xq, const void* dy_ps, float* dw_kn, int accumulate, void* workspace, int V, int HP,
                         int WP, int OH, int OW, int Cout, int KH, int KWP, int stride, simclr_stream_t stream);

/* ---- BatchNorm: tf2/resnet.py:31-78 (BatchNormRelu), residual tail :382/:487 ------------------- */
int simclr_bn_reduce_slots(const float* partial, int nslot, int C, double* sums, simclr_stream_t stream);
/* slots of simclr_conv2d_fwd_pivoted -> raw moments: sums[c] = S1 + count p, sums[C + c] = S2 + 2 p S1 + count p^2 (fp64) */
int simclr_bn_reduce_slots_pivoted(const float* partial, int nslot, int C, const float* pivot, double count, double* sums,
                                   simclr_stream_t stream);
/* BatchNorm statistics of c = h W (1x1 convolution, tf2/resnet.py:470-474 conv3 -> bn3) without forming c:
 * sums[0][n] = sum_k colsum(h)[k] W[k][n], sums[1][n] = sum_k GW[k][n] W[k][n] with GW = (h^T h) W [K][N] (fp32, from
 * simclr_conv2d_gram + simclr_small_gemm_nt_f32), w_kn = the weights as multiplied, fp32 [K][N]; colsum(h) as fp64
 * (cs64) or fp32 (cs32), exactly one non-NULL.  sums: double [2][N], the operand of simclr_bn_finalize. */
int simclr_bn_sums_from_gram(const float* gw, const float* w_kn, const double* cs64, const float* cs32, int K, int N,
                             double* sums, simclr_stream_t stream);
/* sums [2][C] fp64 (after the cross-replica all-reduce) OR partial [nslot][2][C] fp32 (single replica). */
int simclr_bn_finalize(const double* sums, const float* partial, int nslot, double count, int C,
                       const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                       float decay, float eps, float* mean, float* rstd, float* scale, float* shift,
                       simclr_stream_t stream);
/* y = [relu]( x*scale+shift [+ res | + res*rscale+rshift] ).  relu_bits (nullable): receives the ReLU mask
 * (y > 0) as one byte per 16-byte chunk of y (bit e = element e of the chunk: 8 channels bf16, 4 channels f32),
 * i.e. unsigned char [rows][C/8] (bf16) or [rows][C/4] (f32) -- what simclr_conv2d_dgrad_bn mask_mode 3 reads
 * instead of the whole block output. */
int simclr_bn_apply(const void* x, const float* scale, const float* shift, const void* res,
                    const float* rscale, const float* rshift, void* y, unsigned char* relu_bits,
                    long long rows, int C, int relu, int dtype, simclr_stream_t stream);
/* mask_mode: 0 none, 1 (mask_src > 0), 2 (x*scale+shift > 0)  -- the ReLU gradient */
int simclr_bn_bwd_reduce(const void* dy, const void* x, const void* mask_src, const float* scale,
                         const float* shift, const float* mean, const float* rstd, long long rows, int C,
                         int mask_mode, float* partial, int nslot, int dtype, simclr_stream_t stream);
int simclr_bn_bwd_finalize(const double* local_sums, const double* global_sums, const float* partial,
                           int nslot, double count, int C, float* dgamma, float* dbeta, int accumulate,
                           float* c1, float* c2, simclr_stream_t stream);
int simclr_bn_bwd_apply(const void* dy, const void* x, const void* mask_src, const float* scale,
                        const float* shift, const float* mean, const float* rstd, const float* c1,
                        const float* c2, long long rows, int C, int mask_mode, void* dx, void* dmasked,
                        int dtype, simclr_stream_t stream);

/* ---- view packing / pooling: tf2/model.py:250-259, tf2/resnet.py:602-611, :693-696 ------------- */
int simclr_pack_views(const float* images, void* xp, int b, int H, int W, int k, int HP, int WP, int pad,
                      int dtype, simclr_stream_t stream);
/* the same with the pre-split copy of the packed pixels (simclr_presplit_packed) written in the same pass; fp32 only */
int simclr_pack_views_ps(const float* images, void* xp, void* xq, int b, int H, int W, int k, int HP, int WP, int pad,
                         simclr_stream_t stream);
int simclr_bnrelu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y,
                              unsigned char* arg, int V, int H, int W, int C, int OH, int OW, int ksz,
                              int stride, int pad_t, int pad_l, int dtype, simclr_stream_t stream);
int simclr_maxpool_bwd(const void* dy, const unsigned char* arg, void* dx, int V, int H, int W, int C,
                       int OH, int OW, int ksz, int stride, int pad_t, int pad_l, int dtype,
                       simclr_stream_t stream);
/* Stem backward with the max-pool backward fused in (the gradient wrt the BN+ReLU output is never written): BN-backward
 * reduce and apply straight from the pooled gradient + arg-max taps (tf2/resnet.py:602-611 under tape.gradient). */
int simclr_bn_bwd_pool_slots(long long rows, int C, int dtype);
int simclr_bn_bwd_reduce_pool(const void* dy, const unsigned char* arg, const void* x, const float* scale,
                              const float* shift, const float* mean, const float* rstd, int V, int H, int W, int C,
                              int OH, int OW, int ksz, int stride, int pad_t, int pad_l, float* partial, int nslot,
                              int dtype, simclr_stream_t stream);
int simclr_bn_bwd_apply_pool(const void* dy, const unsigned char* arg, const void* x, const float* scale,
                             const float* shift, const float* mean, const float* rstd, const float* c1, const float* c2,
                             void* dx, int V, int H, int W, int C, int OH, int OW, int ksz, int stride, int pad_t, int pad_l,
                             int dtype, simclr_stream_t stream);
int simclr_global_avgpool_fwd(const void* x, void* y, int V, int HW, int C, int dtype,
                              simclr_stream_t stream);
int simclr_global_avgpool_fwd_f32(const void* x, float* y32, int V, int HW, int C, int dtype,
                                  simclr_stream_t stream);   /* same mean, float32 output (input of fp32 heads) */
int simclr_global_avgpool_bwd(const void* dy, const void* mask_src, void* dx, int V, int HW, int C,
                              int dtype, simclr_stream_t stream);

/* ---- on-device augmentation: batch_random_blur, tf2/data_util.py:323-361,413-440 (tf2/model.py:255-258) ---- */
int simclr_batch_blur(const float* images, float* tmp, float* out, const float* filt, const float* selector,
                      int b, int H, int W, int nviews, int K, simclr_stream_t stream);

/* ---- ResNet-D shortcut pool (tf2/resnet.py:330-338,400-408) and selective-kernel unit (:266-277) ---- */
int simclr_avgpool2_fwd(const void* x, void* y, int V, int H, int W, int C, int stride, int dtype,
                        simclr_stream_t stream);
int simclr_avgpool2_bwd(const void* dy, void* dx, int V, int H, int W, int C, int stride, int dtype,
                        simclr_stream_t stream);
int simclr_sk_pool_fwd(const void* a, void* g, int V, int HW, int f, int gpitch, int dtype, simclr_stream_t stream);
int simclr_sk_mix_fwd(const void* a, const void* l, void* out, int V, int HW, int f, int lpitch, int dtype,
                      simclr_stream_t stream);
int simclr_sk_mix_bwd_logits(const void* a, const void* l, const void* dout, void* dl, int V, int HW, int f,
                             int lpitch, int dtype, simclr_stream_t stream);
int simclr_sk_mix_bwd_streams(const void* l, const void* dout, const void* dg, void* da, int V, int HW, int f,
                              int lpitch, int gpitch, int dtype, simclr_stream_t stream);

/* ---- supervised (linear-eval) head tail: tf2/objective.py:27-32, tf2/metrics.py:49-55 ---------- */
int simclr_bias_softmax_xent(const void* z, const float* bias, const int* labels, int rows,
                             int label_rows, int nclass, int cpad, float gscale, void* dlogits,
                             float* out, int dtype, simclr_stream_t stream);
int simclr_colsum(const void* x, int rows, int C, int cvalid, float* out, int accumulate, int dtype,
                  simclr_stream_t stream);

/* ---- small helpers ------------------------------------------------------------------------------ */
/* dtype hand-over between the fp32 heads / loss and the T-typed encoder (tf2/model.py:262-266) */
int simclr_cast(const void* x, void* y, long long n, int dtype_in, int dtype_out, simclr_stream_t stream);
/* y += a*x: gradient of the supervised head's L2 term (tf2/model.py:49-60, run.py:609-612) */
int simclr_axpy_f32(float a, const float* x, float* y, long long n, simclr_stream_t stream);
/* Metric bookkeeping of one step in one launch (tf2/run.py:587-613: update_pretrain_metrics_train, update_finetune_metrics_train,
 * weight_decay, total_loss = the sum of the loss terms).  src: HOST array of n <= 16 device scalar pointers, scale: host array
 * of n factors (NULL = 1): dst[i] += scale[i] * *src[i] (dst NULL: skipped); total (nullable) receives the sum of the scaled
 * terms whose bit is set in total_mask; copy (nullable, n floats) receives the scaled terms themselves -- the step's
 * scalars live in a per-step arena that the next step reuses, the copy is what a caller may keep. */
int simclr_accumulate_scalars(const float* const* src, const float* scale, int n, float* dst, float* total,
                              int total_mask, float* copy, simclr_stream_t stream);
int simclr_l2_loss_f32(const float* x, long long n, float* out, simclr_stream_t stream); /* tf.nn.l2_loss, model.py:49-60 */

/* ---- BatchNorm backward FOLDED into the convolution that produced the BN's input (1x1 expand convs, K <= N): what
 * tape.gradient (tf2/run.py:621) yields for conv -> BatchNormalization (tf2/resnet.py:460-467, the block's last conv + BN)
 * without ever forming the gradient wrt the conv output.  With c = h W, dh = a*dm + b*c + d per channel:
 *   dW = (h^T dm)*a + ((h^T h) W)*b + colsum(h) (x) d ;   d(h) = dm (a*W)^T + h (W diag(b) W^T) + W d.
 * simclr_bn_fold_pre -> a, b, d, W*b (fp32), the first N columns of the extended dgrad weights and the bias W d
 * (simclr_bn_fold_coeffs: a, b, d alone);  [GEMMs h^T dm, h^T h via simclr_conv2d_wgrad; (h^T h) W and (W*b) W^T via simclr_conv2d_fwd];
 * simclr_bn_fold_post -> dW and the last K columns of the extended weights;  simclr_conv2d_dgrad_bn_ext -> d(h) with the fused
 * BN-backward reduce of the producer BN, reading dm and h (K-extended reduction) instead of a materialised dh. ---- */
int simclr_bn_fold_coeffs(const float* scale, const float* mean, const float* rstd, const float* c1, const float* c2,
                          float* a, float* b, float* d, int C, simclr_stream_t stream);
int simclr_bn_fold_pre(const void* w, const float* scale, const float* mean, const float* rstd, const float* c1,
                       const float* c2, float* a, float* b, float* d, float* wb, void* wext, float* e, int K, int N,
                       int dtype, simclr_stream_t stream);
int simclr_bn_fold_post(const float* t1, const float* gw, const double* cs, const float* cs32, const float* a, const float* b,
                        const float* d, const float* q, float* dw, void* wext, int K, int N, int accumulate, int dtype,
                        simclr_stream_t stream);
int simclr_conv2d_dgrad_ext(const void* dm, const void* h, const void* w_ext, const float* bias, void* dx, int accumulate,
                            int V, int H, int W, int Cin, int Cout, int dtype, simclr_stream_t stream);  /* plain epilogue */
/* sum(dm * x^) of the folded BatchNorm from t1 = h^T dm (no pass over the conv output): sums [2][N] fp64, sums[0] = sum dm given */
int simclr_bn_fold_s2(const float* t1, const void* w, const float* mean, const float* rstd, double* sums, int K, int N,
                      int dtype, simclr_stream_t stream);
/* h^T h [K*K] followed by colsum(h) [K] for h [M][K] (T), K in {64, 128, 256(bf16)}: the activation is streamed once.
 * fp32 storage: dtype may carry SIMCLR_FMT_TERMS of the forward pass the statistics serve -- exact arithmetic (or no field and an exact
 * process default) = exact fp32 MFMA; any split mode = six bf16-piece terms (fp32-level products), column sums by exact fp32 MFMA */
size_t simclr_conv2d_gram_workspace_bytes(long long M, int K, int dtype);
int simclr_conv2d_gram(const void* h, float* out, void* workspace, long long M, int K, int dtype, simclr_stream_t stream);
/* C [M][N] = A [M][K] B[N][K]^T, fp32 on the matrix cores (the small K x K / K x N products of the folded form);
 * M, N multiples of 16, K of 32 */
int simclr_small_gemm_nt_f32(const float* A, const float* B, float* C, int M, int N, int K, simclr_stream_t stream);
int simclr_conv2d_dgrad_bn_ext(const void* dm, const void* h, const void* w_ext, const float* bias, void* dx,
                               int accumulate, const void* bn_x, const void* bn_mask, const float* bn_scale,
                               const float* bn_shift, const float* bn_mean, const float* bn_rstd, int mask_mode,
                               float* stats, int nslot, int V, int H, int W, int Cin, int Cout, int dtype,
                               simclr_stream_t stream);

/* ---- two-view training augmentation on the device: tf2/data_util.py:443-475 (preprocess_for_train: random-resized-crop
 * with bicubic resize :246-320/:362-377, random flip :463, colour jitter in random order :54-173/:380-389, random grayscale
 * :48-52, clip :473-474) for every image and both views of tf2/data.py:52-62.  The random draws come from the caller as
 * params [b][views][16] = {crop_y, crop_x, crop_h, crop_w, flip, jitter_on, perm[4], brightness, contrast, saturation,
 * hue_delta, gray_on, 0}; src [b,Hs,Ws,3] float32 in [0,1] (src_dtype 0) or uint8 (src_dtype 2); out [b,H,W,3*views]
 * float32 in [0,1], the layout Model.__call__ (tf2/model.py:241-259) consumes. ---- */
size_t simclr_augment_workspace_bytes(int b, int views, int H, int W);
int simclr_augment_views(const void* src, int src_dtype, const float* params, void* workspace, float* out, int b,
                         int views, int Hs, int Ws, int H, int W, simclr_stream_t stream);

/* ---- collective C without a collective library: tf2/resnet.py:50-60 (SyncBatchNormalization moment all-reduce) -------
 * One-shot exchange over peer-mapped memory (csrc/comm.hip): every rank owns a mailbox all peers map through hipIpc;
 * an exchange is ONE single-workgroup launch per rank that writes its block into every peer's mailbox, publishes a
 * sequence flag, waits (bounded) for the R flags of its own mailbox and adds the R blocks in rank order (bit-identical
 * on every replica).  The 64-byte handles travel between processes by the caller's means (simclr_amd/comm.py:
 * torch.distributed.all_gather_object).  RCCL remains the fallback (FLAGS / SIMCLR_PEER_STATS) and carries A and B. */
size_t simclr_comm_mailbox_bytes(int world, int max_doubles);
/* allocates (library-owned, uncached where available) + zeroes the mailbox; ipc_handle_64: 64 bytes out; 3 = no hipIpc */
int simclr_comm_create(int world, int max_doubles, void** mailbox, void* ipc_handle_64);
int simclr_comm_open(const void* ipc_handle_64, void** mapped);
int simclr_comm_close(void* mapped);
/* Wall-clock bound (seconds) of the arrival wait of the exchanges launched after this call; <= 0 restores the default (600 s or
 * SIMCLR_PEER_STATS_TIMEOUT_S).  Process-wide; used for the short-bounded set-up self-test. */
int simclr_comm_set_timeout(double seconds);
int simclr_comm_destroy(void* mailbox);
/* out[i] = sum_r in_r[i] (rank order), count <= max_doubles fp64 values; peers: HOST array of `world` mapped mailboxes
 * (peers[rank] = own); seq = 1, 2, ... identical on all ranks per exchange; status (nullable device int, zeroed once by the
 * caller) = STICKY count of peer arrivals that timed out (never cleared by the library); an exchange that timed out returns NaN
 * in `out`.  Refuses a capturing stream (seq is a kernel argument: a hipGraph replay would carry a stale sequence number). */
int simclr_comm_stats_allreduce(const double* in, double* out, int count, void* const* peers, int rank, int world,
                                int max_doubles, unsigned seq, int* status, simclr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMCLR_HIP_H_ */
