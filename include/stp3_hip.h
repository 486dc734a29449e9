/*
 * stp3_hip.h -- C ABI of libstp3hip.so: the MI355X (gfx950) implementation of ST-P3's
 * LSS camera->BEV lifting hot path.
 *
 * Boundary.  The reference is pure Python; the operator boundary this library replaces is
 *   - stp3/utils/geometry.py:299-330   VoxelsSumming.apply(x, geometry, ranks)
 * and, one level up, the code that feeds it:
 *   - stp3/models/stp3.py:186-201      STP3.get_geometry
 *   - stp3/models/stp3.py:215-221      softmax(depth) (x) feature outer product
 *   - stp3/models/stp3.py:226-301      STP3.projection_to_birds_eye_view (ego alignment,
 *                                      voxel index, sort, per-voxel sum, discounted accumulate)
 * The Python host (st-p3_amd/stp3_amd/ops.py) binds these entry points with ctypes and wraps
 * them in torch.autograd.Functions; INTEGRATION.md shows the stub a maintainer of the reference
 * would add.
 *
 * Conventions
 *   - every pointer except `dims` is a DEVICE pointer into caller-owned memory that must stay
 *     valid until the work enqueued on `stream` has completed; the library allocates nothing;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only enqueue,
 *     they never synchronise, so they can be captured into a hipGraph;
 *   - return value: 0 on success, a negative hipError_t on a launch failure, or one of the
 *     STP3_E* codes for rejected arguments.  Nothing throws, nothing exits;
 *   - re-entrant: no global mutable state, no environment variables: every choice is an argument.
 *
 * Layouts ("pixel-major" = channels-last memory of the corresponding NCHW tensor)
 *   pix   = (n*fH + h)*fW + w                      camera pixel index inside one (b,t) frame
 *   feat  [B*T][N*fH*fW][C]     float32            encoder features   (stp3.py:208, x)
 *   depth [B*T][N*fH*fW][D]     float32            depth logits (and their gradient)
 *   prob_cm / vox_cm [B*T][N*fW][D][fH]            depth probabilities / voxel ids, column-major: (camera, column),
 *                                                  depth bin, image row
 *   vox   int32 voxel id = ix*(Y*Z) + iy*Z + iz, or -1 if the point falls outside the grid
 *   bev   [B][T][C][X][Y]       float32            reference layout (stp3.py:230-232)
 */
#ifndef STP3_HIP_H
#define STP3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STP3_OK          0
#define STP3_EINVAL   (-10001)  /* bad dimension / null pointer */
#define STP3_EUNSUP   (-10002)  /* valid request this build does not support (e.g. Z != 1 for pooling) */
#define STP3_ENOSPACE (-10003)  /* workspace too small */

/* element types of activation / gradient tensors */
#define STP3_DTYPE_F32  0
#define STP3_DTYPE_BF16 1

/* Problem shape.  P = N*D*fH*fW frustum points per (b,t); V = X*Y*Z voxels. */
typedef struct stp3_lift_dims {
    int32_t B, T, N;      /* batch, frames (time receptive field), cameras           */
    int32_t D, fH, fW;    /* depth bins, feature-map height/width                     */
    int32_t C;            /* feature channels (cfg.MODEL.ENCODER.OUT_CHANNELS)        */
    int32_t X, Y, Z;      /* BEV grid (geometry.py:56-57)                             */
} stp3_lift_dims;

/* order of the voxel-id array written by stp3_voxel_index */
#define STP3_VOX_REFERENCE  0   /* [B*T][N][D][fH][fW] -- the reference's flatten order (stp3.py:284) */
#define STP3_VOX_PIXELMAJOR 1   /* [B*T][N*fH*fW][D]   -- what the pooling kernels consume           */

/* library / build identification: returns a static string such as "stp3hip 0.1 gfx950" */
const char* stp3_version(void);

/*
 * stp3_voxel_index -- frustum point -> voxel id, bit-exact with the reference's CPU arithmetic.
 * Replaces stp3.py:192-198 (get_geometry), :270-277 (ego alignment), :287-289 (index),
 * :239-255 (range mask + rank).  Float32, one rounding per operation, left-to-right
 * ((m0*x + m1*y) + m2*z) + t, true division, truncation toward zero.
 *
 *   cam_m  [B*T*N][9]  R . K^-1 per camera (row-major 3x3), built by the host with torch CPU ops
 *   cam_t  [B*T*N][3]  camera translation in the ego frame
 *   ego_r  [B*T][9], ego_t [B*T][3]   pose_vec2mat(future_egomotion) (geometry.py:158-172);
 *          frame k of sample b is mapped by ego[b][k], ego[b][k+1], ..., ego[b][T-2] in turn
 *   xs [fW], ys [fH], ds [D]          the separable frustum (stp3.py:111-130)
 *   bev_offset [3] = bev_start - bev_res/2 (float32), bev_res [3]
 *   vox  [B*T*P] int32 out, in `order`
 *   counts [B*T*V] int32 or NULL: if non-NULL it must be zero-filled by the caller; the kernel
 *          adds the number of points per voxel (a diagnostic histogram; the plan does not need it).
 */
int stp3_voxel_index(const stp3_lift_dims* dims,
                     const float* cam_m, const float* cam_t,
                     const float* ego_r, const float* ego_t,
                     const float* xs, const float* ys, const float* ds,
                     const float* bev_offset, const float* bev_res,
                     int order, int32_t* vox, int32_t* counts, void* stream);

/*
 * Pooling plan: the geometry-only structure that the forward kernel consumes.  It replaces the
 * reference's get_geometry + ego alignment + index + boolean mask + argsort (stp3.py:192-198,
 * :270-277, :287-289, :239-257) and depends only on the camera / ego poses, so it can be built on a
 * side stream while the image encoder is running.
 *
 * Along an image column (camera n, feature column w, depth bin d) consecutive rows h fall into
 * the same BEV cell most of the time; a RUN is a maximal set of consecutive h with one voxel
 * id >= 0.  The forward pass sums every run once (pass 1, per image column) into the run's SLOT and
 * then every voxel over its runs (pass 2); the plan numbers the runs (slot = position in the
 * enumeration frame, column, last row, depth bin) and lists them per voxel in ascending order, so
 * that every voxel is summed in ONE fixed order.
 *
 *   stp3_lift_plan_bytes : size of the plan buffer for `dims`
 *   stp3_lift_plan_build : geometry inputs exactly as for stp3_voxel_index; writes
 *                            vox_cm [B*T][N*fW][D][fH] int32  voxel ids in COLUMN-MAJOR order (the same ids
 *                                                            stp3_voxel_index computes; the rows h of an image column
 *                                                            and depth bin are contiguous -- the order the backward
 *                                                            kernel walks them in)
 *                            plan   sections, each 256-byte aligned, in this order:
 *                                   vox_off  [B*T][V+1] int32   exclusive scan of runs per voxel, per frame
 *                                   masks    [B*T][N*fW][fH] 2 x uint64: bit d of word 0 = a run of depth bin d ENDS
 *                                            at this row, bit d of word 1 = point (d, h) lies inside the BEV grid
 *                                   col_cnt  [B*T*N*fW] int32   runs per column (scratch)
 *                                   col_off  [B*T*N*fW + 1] int32  exclusive scan over all frames: slot of the
 *                                            column's first run; col_off[bt*N*fW] = first slot of frame bt; the
 *                                            last entry = total number of runs
 *                                   tmp      [B*T*P] int32      (scratch)
 *                                   vox_runs [B*T*P] int32      slots of voxel v of frame bt, ascending, at
 *                                            col_off[bt*N*fW] + vox_off[bt][v] .. + vox_off[bt][v+1]
 *                                   run_desc [B*T*P] uint32     per slot: depth bin | first row << 8 | last row << 16
 *                                   run_vox  [B*T*P] int32      per slot: the run's voxel
 *                          counts = int32 [B*T][V] scratch that must be ZERO on entry (the caller zero-fills
 *                          it once; every build leaves it zero again).
 *   Launches: columns (ids, masks, counts), scan of the columns, scan of the voxels, fill, per-voxel order.
 *   Limits: Z == 1, C % 4 == 0, C <= 64, D <= 64, fH <= 128, N*fW < 4096, B*T*N*fH*fW*max(C,D)*4 < 2^32,
 *   else STP3_EUNSUP.
 */
int stp3_lift_plan_bytes(const stp3_lift_dims* dims, size_t* bytes);
int stp3_lift_plan_build(const stp3_lift_dims* dims,
                         const float* cam_m, const float* cam_t,
                         const float* ego_r, const float* ego_t,
                         const float* xs, const float* ys, const float* ds,
                         const float* bev_offset, const float* bev_res,
                         int32_t* vox_cm, int32_t* counts, void* plan, size_t plan_bytes, void* stream);

/* stp3_depth_softmax -- softmax over the D depth bins of every pixel (stp3.py:215).
 * logits [B*T][N*fH*fW][D] float32 (pixel-major) -> prob_cm [B*T][N*fW][D][fH] float32 (column-major; must not
 * alias logits).  stp3_lift_splat_fwd computes the same values on the fly; this entry point is the stand-alone
 * operator. */
int stp3_depth_softmax(const stp3_lift_dims* dims, const float* logits, float* prob, void* stream);

/* memory layout of the BEV tensor handed to / produced by the pooling calls */
#define STP3_BEV_CHANNELS_FIRST 0   /* [B][T][C][X][Y]  the reference's layout (stp3.py:230-232, :297-299)          */
#define STP3_BEV_CHANNELS_LAST  1   /* [B][T][X][Y][C]  = channels-last memory of the same (B,T,C,X,Y) tensor: what the */
                                    /* NHWC convolutions of the temporal model consume; no transpose pass             */
#define STP3_BEV_CHANNELS_LAST_BF16 2 /* (forward output only) the same layout, every value rounded ONCE to bf16: what  */
                                    /* a bf16 temporal model makes of the float32 BEV in its first operator anyway --  */
                                    /* half the bytes written and no cast pass in the consumer                         */

/*
 * stp3_lift_splat_fwd -- out[b][t] = sum_{k<=t} discount^(t-k) Pool_k,
 *   Pool_k[c][v] = sum over points p of frame k with vox(p) == v of softmax_D(logits)[p] * feat[pix(p)][c].
 * Replaces stp3.py:215 (depth softmax), :216-221 (outer product, never materialised), geometry.py:302-318
 * (VoxelsSumming.forward) and stp3.py:279-299 (scatter, discount, permute).  Two kernels:
 *   pass 1  per image column; the column's logits and features go global -> LDS directly and every feature and logit
 *           is read from memory exactly once.  Columns of at most 32 rows with C == 64 (lift_column_mma_kernel): ONE
 *           WORKGROUP of four waves per column -- the waves share the staged column, split the rows of the softmax
 *           (in place in LDS) and deal out the 32-run tiles; the run sums are a matrix product  masked probabilities
 *           [runs x rows] x features [rows x C]  on the matrix cores (v_mfma_f32_32x32x2_f32), 32 runs = 32 slots per
 *           tile; this path does NOT write prob_cm (its backward recomputes the probabilities from the logits).
 *           Other shapes (lift_column_kernel): one wave per column, lane = channel, ONE walk over the rows with an
 *           accumulator per depth bin (v_fmac_f32 with a DPP row broadcast of the probability); a finished run's
 *           C-vector goes to its slot; the probabilities are written to prob_cm for the general backward
 *   pass 2  16 lanes per voxel add the voxel's slots (ascending) and carry the discounted state through the T
 *           frames in registers; every BEV row (256 bytes) is written once
 * Deterministic (fixed summation order, no atomics).
 *   feat    [B*T][N*fH*fW][C] float32, logits [B*T][N*fH*fW][D] float32 (pixel-major)
 *   prob_cm [B*T][N*fW][D][fH] float32 out, or NULL (needed by the backward only when stp3_lift_bwd_needs_prob says so)
 *   workspace  stp3_lift_workspace_bytes(dims): one slot per possible run (B*T*P*C floats -- the geometry decides how
 *              many are touched: ~7 % for nuScenes-like rigs) + one BEV-sized buffer
 *   bev_layout STP3_BEV_CHANNELS_LAST : pass 2 writes bev directly
 *              STP3_BEV_CHANNELS_FIRST: pass 2 writes into the workspace and a transpose pass produces bev
 *   bev: B*T*C*X*Y float32, fully overwritten (empty voxels get 0).
 */
int stp3_lift_workspace_bytes(const stp3_lift_dims* dims, size_t* bytes);
int stp3_lift_splat_fwd(const stp3_lift_dims* dims, const float* feat, const float* logits,
                        const void* plan, float discount, int bev_layout,
                        void* workspace, size_t workspace_bytes, float* prob_cm, void* bev, void* stream);

/*
 * stp3_lift_splat_bwd -- gradients of stp3_lift_splat_fwd (softmax included).
 * Replaces autograd through stp3.py:215-301 and VoxelsSumming.backward (geometry.py:320-330).  No atomics.
 *   grad_bev   in `bev_layout` and `grad_dtype` (channels-last float32 / bfloat16, channels-first float32)
 * Columns of at most 32 rows with C == 64 (stp3_lift_bwd_needs_prob -> 0): ONE kernel, one workgroup per image column,
 * the adjoint of the forward's matrix product on the matrix cores,
 *       dM[run][h] = sum_c G[run][c] feat[h][c],    dfeat[h][c] = sum_run M[run][h] G[run][c],
 *   G[run] = the gradient row of the run's voxel, G_t = sum_{t' >= t} discount^(t'-t) grad_bev[b][t'] summed on the fly
 *   over the frames of a channels-last gradient (a channels-first gradient goes through an import pass that transposes
 *   and sums it into `workspace`); probabilities recomputed from `logits` exactly as in the forward; softmax backward
 *   fused.  Reads `logits` and `plan`; prob_cm / vox_cm are not used and may be NULL.
 * Other shapes (stp3_lift_bwd_needs_prob -> 1): import pass (layout / dtype conversion carrying the recurrence,
 *   voxel-major [B*T][V][C] float32 into `workspace`) + a gather kernel, one image pixel per lane pair, walking the depth
 *   bins 8 at a time with the run table and gradient rows staged in LDS.  Reads prob_cm (written by the forward) and
 *   vox_cm; logits / plan are not used and may be NULL.
 *   workspace  stp3_lift_workspace_bytes(dims), always required
 *   grad_feat  [B*T][N*fH*fW][C], grad_logits [B*T][N*fH*fW][D]   outputs, fully overwritten
 */
int stp3_lift_bwd_needs_prob(const stp3_lift_dims* dims, int* needs);
int stp3_lift_splat_bwd(const stp3_lift_dims* dims, const void* grad_bev, int bev_layout, int grad_dtype,
                        const float* feat, const float* logits, const float* prob_cm, const int32_t* vox_cm,
                        const void* plan, float discount,
                        void* workspace, size_t workspace_bytes,
                        float* grad_feat, float* grad_logits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise 2-D convolution of the EfficientNet trunk (MBConv blocks), channels-last.
 * Replaces the groups == channels Conv2dStaticSamePadding calls that the reference's Encoder drives
 * (stp3/models/encoder.py:62-70 -> efficientnet_pytorch MBConvBlock._depthwise_conv; forward and the
 * two gradients autograd derives from it).
 *   x  [N][H][W][C], y / dy [N][Ho][Wo][C]   dtype = STP3_DTYPE_F32 or STP3_DTYPE_BF16 (C % 4 / C % 8 == 0)
 *   w  [K*K][C] float32 (tap-major), dw same layout, float32
 *   K in {3,5}, stride in {1,2}; pad_top / pad_left = leading zero padding (the trailing padding is
 *   implied by Ho, Wo: "static same" padding is asymmetric on stride-2 layers)
 *   K = 7 at stride 1: the 7x7 depthwise layer of the ConvNeXt blocks of the prediction stage
 *   (stp3/layers/convolutions.py:309-345 `Block.dwconv`, nn.Conv2d(dim, dim, 7, padding=3, groups=dim) WITH a bias:
 *   stp3_dwconv2d_fwd_bias adds bias[C] float32 (NULL: none); its gradient is the per-channel sum of dy)
 * bwd_weight is deterministic (two-stage reduction through `workspace`, no atomics).
 */
typedef struct stp3_dwconv_dims {
    int32_t N, H, W, C;          /* input  (channels-last)            */
    int32_t Ho, Wo;              /* output spatial size               */
    int32_t K, stride;           /* square kernel, stride             */
    int32_t pad_top, pad_left;   /* leading zero padding              */
    int32_t dtype;               /* STP3_DTYPE_*  of x / y / dy / dx  */
} stp3_dwconv_dims;

int stp3_dwconv2d_fwd(const stp3_dwconv_dims* dims, const void* x, const float* w, void* y, void* stream);
int stp3_dwconv2d_fwd_bias(const stp3_dwconv_dims* dims, const void* x, const float* w, const float* bias, void* y,
                           void* stream);
int stp3_dwconv2d_bwd_data(const stp3_dwconv_dims* dims, const void* dy, const float* w, void* dx, void* stream);
int stp3_dwconv2d_bwd_weight_workspace(const stp3_dwconv_dims* dims, size_t* bytes);
int stp3_dwconv2d_bwd_weight(const stp3_dwconv_dims* dims, const void* x, const void* dy, float* dw,
                             void* workspace, size_t workspace_bytes, void* stream);
/* the same with dw in the layout of the nn.Conv2d parameter, [C][1][K][K] (what autograd hands to the optimizer as it is;
 * stp3_dwconv2d_bwd_weight writes the tap-major [K*K][C] the kernels read their weights in) */
int stp3_dwconv2d_bwd_weight_oihw(const stp3_dwconv_dims* dims, const void* x, const void* dy, float* dw,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Element-wise half of the convolutional GRU cells of the prediction stage.  Replaces, around
 * the three convolutions of `SpatialGRU.gru_cell` (stp3/layers/temporal.py:42-56) and `Dual_GRU.gru_cell_1 / _2`
 * (:118-145):  update, reset = sigmoid(conv([x, state]) + bias_init);  tilde = conv([x, (1 - reset) * state]);
 * out = (1 - update) * state + update * tilde  -- and the gradients autograd derives from them.
 *   rows = N * H * W pixels, channels-last.  xs, xs2, dxs2, acc [rows][Cx + C] = [x | state];
 *   gates, dgates [rows][2C] = [update | reset] PRE-activations (the two gate convolutions run as one with 2C outputs);
 *   tilde, dtilde, out [rows][C];  dout [rows][ld_dout >= C].  dtype STP3_DTYPE_F32 / _BF16 (float32 arithmetic, one
 *   rounding per result); Cx, C multiples of 16 bytes of elements, 16-byte aligned pointers (STP3_EUNSUP otherwise).
 *   reset_cat_fwd : xs2 = [x | (1 - r) state]                       (the operand of the tilde convolution)
 *   output_fwd    : out = (1 - u) state + u tilde
 *   output_bwd    : dtilde = dout u;  dgates[:, :C] = dout (tilde - state) u (1 - u)
 *   reset_cat_bwd : acc = [dxs2_x | dout (1 - u) + dxs2_state (1 - r)];  dgates[:, C:] = -dxs2_state state r (1 - r)
 *                   (the caller adds the data gradient of the gate convolution to acc: d[x | state])
 */
typedef struct stp3_gru_dims {
    int64_t rows;
    int32_t Cx, C;               /* channels of x and of the state              */
    int32_t dtype;               /* STP3_DTYPE_*                                */
    float bias_init;             /* gru_bias_init of the reference (default 0)  */
} stp3_gru_dims;

int stp3_gru_reset_cat_fwd(const stp3_gru_dims* dims, const void* xs, const void* gates, void* xs2, void* stream);
int stp3_gru_output_fwd(const stp3_gru_dims* dims, const void* gates, const void* xs, const void* tilde, void* out,
                        void* stream);
int stp3_gru_output_bwd(const stp3_gru_dims* dims, const void* dout, int32_t ld_dout, const void* gates, const void* xs,
                        const void* tilde, void* dtilde, void* dgates, void* stream);
int stp3_gru_reset_cat_bwd(const stp3_gru_dims* dims, const void* dout, int32_t ld_dout, const void* gates, const void* xs,
                           const void* dxs2, void* acc, void* dgates, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the channels of every pixel (+ GELU), channels-last rows -- the prediction stage (SURVEY.md section 8 row
 * f2).  Replaces `LayerNorm` of stp3/layers/convolutions.py:283-307 (F.layer_norm over C in its channels_last form, the
 * hand-written mean / variance over dim 1 in its channels_first form: the same normalisation) and the nn.GELU() that
 * follows it in `Bottleblock` (:347-380); `Block` (:309-345) uses it without the activation.
 *   x, dx [rows][ldx >= C], y / dy [rows][ldy >= C]   dtype STP3_DTYPE_F32 / _BF16, float32 arithmetic, one rounding
 *   y = act((x - mean_c) / sqrt(var_c + eps) * gamma + beta), biased variance; gamma / beta [C] float32 (NULL: 1 / 0)
 *   act = STP3_ACT_NONE or STP3_ACT_GELU (defined with the other activations below; exact: 0.5 v (1 + erf(v / sqrt 2)))
 *   C / (16 bytes of elements) must be a power of two in 4 .. 64, ld multiples of the vector, 16-byte aligned pointers (STP3_EUNSUP
 *   otherwise: the caller keeps torch's operator)
 * bwd recomputes the row statistics from x; dgamma / dbeta [C] float32 (either may be NULL) through a two-stage
 * deterministic reduction in `workspace` (stp3_layernorm_bwd_workspace bytes).
 */
typedef struct stp3_layernorm_dims {
    int64_t rows;                /* pixels: N * H * W                         */
    int32_t C, ldx, ldy;         /* channels, row strides (elements)          */
    int32_t dtype;               /* STP3_DTYPE_*                              */
    int32_t act;                 /* STP3_ACT_NONE | STP3_ACT_GELU             */
    float eps;
} stp3_layernorm_dims;

int stp3_layernorm_fwd(const stp3_layernorm_dims* dims, const void* x, const float* gamma, const float* beta, void* y,
                       void* stream);
int stp3_layernorm_bwd_workspace(const stp3_layernorm_dims* dims, size_t* bytes);
int stp3_layernorm_bwd(const stp3_layernorm_dims* dims, const void* dy, const void* x, const float* gamma,
                       const float* beta, void* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused BatchNorm (+ per-sample bias) + activation (+ residual), channels-last, forward / backward.
 * Replaces the nn.BatchNorm2d/3d -> ReLU / swish (-> "+ skip") chains of
 * stp3/layers/convolutions.py:183-280, stp3/layers/temporal.py:252-273,315-325,426-489,
 * stp3/models/decoder.py:22-140 and of the EfficientNet MBConv blocks driven by
 * stp3/models/encoder.py:57-97; the statistics are handed back between the two passes so the host
 * can all-reduce them over RCCL (train.py:47 sync_batchnorm=True).
 *
 *   x, y, res, dy, dx, dres : [N][rows][C] with row strides ldx (x, dx), ldy (y, dy), ldr (res, dres)
 *                             elements, dtype STP3_DTYPE_F32 / _BF16 (16-byte vector path when C, the
 *                             strides and the pointers allow it, scalar path otherwise)
 *   sbias  [N][C] float32    optional per-sample bias added to x before the normalisation (the ASPP
 *                             image-pooling branch and the pyramid-pooling branch are spatially
 *                             constant: convolutions.py:242-270, temporal.py:375-423)
 *   oscale [N]    float32    optional per-sample scale of the activated output (drop-connect)
 *   forward : y = act(BN(x + sbias) [+ res if BEFORE_ACT]) * oscale [+ res if AFTER_ACT]
 *   sums   [2][C] float32    sum and sum of squares of (x + sbias) over the `count` elements per
 *                             channel they cover (N*rows locally; the host may add other ranks' sums)
 *   stp3_bn_apply_fwd : sums != NULL -> training mode (batch statistics; writes save_mean /
 *                       save_invstd [C]; updates running_mean / running_var with `momentum` and the
 *                       unbiased variance when they are non-NULL); sums == NULL -> inference mode
 *                       (running statistics).
 *   stp3_bn_bwd_reduce: sample_sums [N][3][C] and sums [3][C] = sums of g, g * xhat and xhat, where
 *                       g = dy * oscale * act'(.) is the gradient at the BatchNorm output
 *                       (dbeta = sums[0], dgamma = sums[1]; the per-sample sums give the sbias gradient
 *                        gamma*invstd*(Sg_n - rows*sums[0]/count - Sxhat_n*sums[1]/count));
 *                       sample_sums is written only when has_sbias is set -- its one consumer)
 *   stp3_bn_apply_bwd : dx = gamma * invstd * (g - sums[0]/count - xhat * sums[1]/count)
 *                       (sums == NULL: inference-mode BatchNorm, dx = gamma * invstd * g);
 *                       dres (BEFORE_ACT only, may be NULL) = g.  For AFTER_ACT dres is dy itself.
 *   workspace: stp3_bn_workspace_bytes(dims) bytes.  Deterministic (no atomics).
 */
#define STP3_ACT_NONE  0
#define STP3_ACT_RELU  1
#define STP3_ACT_SWISH 2
#define STP3_ACT_GELU  3   /* stp3_layernorm_* only */
#define STP3_RES_NONE       0
#define STP3_RES_BEFORE_ACT 1
#define STP3_RES_AFTER_ACT  2
#define STP3_BN_MAX_ROW_BLOCKS 128

typedef struct stp3_bn_dims {
    int32_t N, rows, C;        /* samples, rows (H*W) per sample, channels          */
    int32_t ldx, ldy, ldr;     /* row strides in elements                            */
    int32_t dtype;             /* STP3_DTYPE_*                                       */
    int32_t act, res_mode;     /* STP3_ACT_*, STP3_RES_*                             */
    int32_t has_sbias, has_oscale;
    int32_t cpad;              /* 0, or the padded channel count of ZERO-PADDED ROWS: C <= cpad <= ldx, ldy (, ldr), cpad a
                                  multiple of 8.  Lanes [C, cpad) of every row of x / res / dy are read as zero whatever
                                  they hold, and the kernels write zeros there in y / dx / dres: a 35-channel layer
                                  lives in 40-lane rows that the MFMA convolutions (Cin, Cout % 8 == 0) consume and
                                  produce in place, and it runs on the 16-byte vector path.  The per-channel arrays
                                  keep C entries.                                                                  */
} stp3_bn_dims;

int stp3_bn_workspace_bytes(const stp3_bn_dims* dims, size_t* bytes);
int stp3_bn_stats(const stp3_bn_dims* dims, const void* x, const float* sbias, void* workspace,
                  size_t workspace_bytes, float* sums, void* stream);
int stp3_bn_apply_fwd(const stp3_bn_dims* dims, const void* x, const float* sbias, const void* res,
                      const float* oscale, const float* sums, double count, const float* gamma,
                      const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* save_mean, float* save_invstd, void* y, void* stream);
int stp3_bn_bwd_reduce(const stp3_bn_dims* dims, const void* dy, const void* x, const float* sbias,
                       const void* res, const float* oscale, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, void* workspace, size_t workspace_bytes,
                       float* sample_sums, float* sums, void* stream);
int stp3_bn_apply_bwd(const stp3_bn_dims* dims, const void* dy, const void* x, const float* sbias,
                      const void* res, const float* oscale, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, const float* sums, double count, void* dx,
                      void* dres, void* stream);
/* Single-process composites (no cross-replica exchange between the passes; same kernels, one call):
 *   stp3_bn_fwd_train : stats + apply.  stat_buf [4][C] float32 receives sum, sum of squares, mean, invstd.
 *   stp3_bn_bwd_train : reduce + apply. dres as in stp3_bn_apply_bwd; mean / invstd from the forward;
 *                       sum_buf [N+1][3][C] float32 receives the per-sample sums and, last, their totals. */
int stp3_bn_fwd_train(const stp3_bn_dims* dims, const void* x, const float* sbias, const void* res,
                      const float* oscale, const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, float* stat_buf, void* workspace,
                      size_t workspace_bytes, void* y, void* stream);
int stp3_bn_bwd_train(const stp3_bn_dims* dims, const void* dy, const void* x, const float* sbias,
                      const void* res, const float* oscale, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, void* workspace, size_t workspace_bytes,
                      float* sum_buf, void* dx, void* dres, void* stream);

/* stp3_bn_dsbias -- gradient of the per-sample bias (the spatially constant branches folded into a BatchNorm: the ASPP
 * image-pooling branch, stp3/layers/convolutions.py:242-270, the pyramid pooling and the ego-motion planes of the
 * temporal blocks, stp3/layers/temporal.py:380-489, stp3/models/stp3.py:145-152) from the sums of stp3_bn_bwd_reduce:
 *   dsbias[n][c] = gamma[c] invstd[c] (S0[n][c] - rows gsums[0][c] / count - S2[n][c] gsums[1][c] / count)
 *   gsums NULL (evaluation mode, running statistics are constants): gamma invstd S0[n][c].
 *   sample_sums [N][3][C], gsums [3][C] (possibly summed over ranks, count = elements over all ranks), dsbias [N][C].
 * stp3_sum_n -- y = src[0] + ... + src[n-1], n <= 8 dense tensors of numel elements (f32 / bf16) in ONE pass, float32
 *   accumulation in input order, one rounding: the gradients of a tensor with several consumers (the reference's
 *   autograd adds them pairwise: stp3/layers/convolutions.py:256-266 ASPP branches, stp3/models/decoder.py:112-126
 *   heads, stp3/layers/temporal.py:470-489 temporal-block paths).  src: HOST array of n device pointers. */
int stp3_bn_dsbias(int32_t N, int32_t C, int32_t rows, const float* sample_sums, const float* gsums, double count,
                   const float* gamma, const float* invstd, float* dsbias, void* stream);
int stp3_sum_n(int32_t n, int64_t numel, int32_t dtype, const void* const* src, void* y, void* stream);
/* stp3_sum_n_plane -- stp3_sum_n plus one addend that is constant over the plane of every (sample, channel): the gradient of a
 *   whole-plane mean (the ASPP image pooling of stp3/layers/convolutions.py:229-240, the pyramid pooling of
 *   stp3/layers/temporal.py:380-424), which torch materialises as a full tensor before adding it.  The dense tensors are
 *   channels-last [N][H*W][C] (per_sample = H*W*C elements per sample); plane [N][C] in the tensors' type; added last. */
int stp3_sum_n_plane(int32_t n, int64_t numel, int32_t dtype, const void* const* src, const void* plane, int64_t per_sample,
                     int32_t C, void* y, void* stream);

/* stp3_linear_fwd / _bwd -- y [M][N] = x [M][K] w[N][K]^T + b[N] (b may be NULL), float32, for the pooled descriptors of the BEV
 *   networks: 1x1 convolutions of maps that are constant over the plane (ASPP image pooling, stp3/layers/convolutions.py:229-240;
 *   pyramid pooling, stp3/layers/temporal.py:380-424; the ego-motion planes, stp3/models/stp3.py:145-152) are products of a few
 *   rows whose time is launch latency -- one launch forward, ONE backward for dx [M][K], dw [N][K] and db [N] (each may be NULL:
 *   not needed).  Deterministic: every sum is split over 16 lanes in a fixed way and finished by a fixed tree.
 *   ldw / lddw: row strides (in floats, >= K) of w and dw -- the weight may be a run of COLUMNS of a wider parameter (the columns
 *   of ASPP's projection that multiply the pooled vector, of a temporal block's 1x1x1 kernel that multiply the ego-motion
 *   planes), read in place, and its gradient written straight into the same columns of the parameter's gradient. */
int stp3_linear_fwd(int32_t M, int32_t K, int32_t N, const float* x, const float* w, int32_t ldw, const float* b, float* y,
                    void* stream);
int stp3_linear_bwd(int32_t M, int32_t K, int32_t N, const float* dy, const float* x, const float* w, int32_t ldw, float* dx,
                    float* dw, int32_t lddw, float* db, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense 2-D convolution, bf16 MFMA implicit GEMM, NHWC (csrc/stp3_conv.hip).
 * Replaces the nn.Conv2d (and frame-folded nn.Conv3d) contractions of stp3/layers/convolutions.py:183-280,
 * stp3/layers/temporal.py:252-273,315-325, stp3/models/decoder.py:22-140 and the stem / 1x1 expand / project
 * convolutions of the MBConv blocks driven by stp3/models/encoder.py:57-97 (groups == 1).
 *   x [N][H][W][ldx >= Cin]  bf16 (row stride ldx lets the kernel read a channel-slice / concat view)
 *   w [Cout][KH][KW][Cin]    bf16 (= channels-last memory of the (Cout, Cin, KH, KW) parameter)
 *   bias [Cout] float32 or NULL
 *   y [N][Ho][Wo][ldy >= Cout]  out_dtype = STP3_DTYPE_BF16 or STP3_DTYPE_F32, float32 accumulation
 *   y[n][ho][wo][co] = bias[co] + sum_{kh,kw,ci} x[n][ho*stride - pad_h + kh*dil_h][wo*stride - pad_w + kw*dil_w][ci]
 *                                                 * w[co][kh][kw][ci]           (zero padding)
 *   sums [2][Cout] float32 or NULL (bf16 outputs only): per-channel sum and sum of squares of the ROUNDED outputs over
 *        all N*Ho*Wo pixels -- the BatchNorm statistics, taken in the epilogue (the result feeds stp3_bn_apply_fwd
 *        directly, no stp3_bn_stats pass over the convolution output); needs `workspace` of
 *        stp3_conv2d_fwd_workspace(dims) bytes (one partial row per 128-pixel workgroup; deterministic)
 * Workgroup tile 128 pixels x 128 (Cout > 64) or 64 output channels, K steps of 64 staged through LDS,
 * v_mfma_f32_32x32x16_bf16.  Requires Cin % 8 == 0, ldx % 8 == 0, 16-byte aligned x / w / y,
 * N*H*W*ldx < 2^31.  The data gradient of a stride-1 convolution is the same call on dy with the taps flipped and
 * Cin / Cout swapped (the host prepares that weight).
 */
typedef struct stp3_conv_dims {
    int32_t N, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t KH, KW, stride;
    int32_t pad_h, pad_w, dil_h, dil_w;
    int32_t ldx, ldy;
    int32_t out_dtype, has_bias;
} stp3_conv_dims;

int stp3_conv2d_fwd_workspace(const stp3_conv_dims* dims, size_t* bytes);
int stp3_conv2d_fwd(const stp3_conv_dims* dims, const void* x, const void* w, const float* bias, void* y,
                    float* sums, void* workspace, size_t workspace_bytes, void* stream);

/* Convolution -> BatchNorm -> activation with the convolution output NEVER in memory (csrc/stp3_conv.hip, epilogue
 * modes): for layers whose convolution is cheap next to the tensor it produces -- the 1x1 expand convolutions of the
 * EfficientNet MBConv blocks driven by stp3/models/encoder.py:57-97 (24..160 -> 144..960 channels: the expanded tensor is
 * 6x the block input) -- the forward runs the convolution twice and the BatchNorm backward recomputes the output tiles
 * from the input instead of reading a stored copy:
 *   stp3_conv2d_fwd_stats     sums [2][Cout] = per-channel sum / sum of squares of the bf16-ROUNDED outputs; nothing stored
 *   stp3_conv2d_fwd_bnact     y = act(scale * e0 + shift), e0 = the rounded convolution output, coef = [scale | shift |
 *                             mean | invstd][Cout] as stp3_bn_finalize writes them
 *   stp3_conv2d_bn_bwd_reduce sums [2][Cout] = sum g, sum g * xhat with g = dz * act'(scale * e0 + shift),
 *                             xhat = (e0 - mean) * invstd; dz [M][ldz] bf16 = the gradient at the activation output
 *   stp3_conv2d_bn_bwd_apply  dy = scale * (g - gsums[0] / count - xhat * gsums[1] / count): the gradient at the convolution
 *                             output (what stp3_bn_apply_bwd writes), gsums possibly summed over ranks
 * dims / x / w as stp3_conv2d_fwd (bf16, no bias, out_dtype bf16; Cout, ldy, ldz multiples of 8); workspace:
 * stp3_conv2d_fwd_workspace.  Same values as the stored route: every mode rounds the accumulators to bf16 first. */
int stp3_conv2d_fwd_stats(const stp3_conv_dims* dims, const void* x, const void* w, float* sums, void* workspace,
                          size_t workspace_bytes, void* stream);
int stp3_conv2d_fwd_bnact(const stp3_conv_dims* dims, const void* x, const void* w, const float* coef, int32_t act, void* y,
                          void* stream);
int stp3_conv2d_bn_bwd_reduce(const stp3_conv_dims* dims, const void* x, const void* w, const void* dz, int32_t ldz,
                              const float* coef, int32_t act, float* sums, void* workspace, size_t workspace_bytes,
                              void* stream);
int stp3_conv2d_bn_bwd_apply(const stp3_conv_dims* dims, const void* x, const void* w, const void* dz, int32_t ldz,
                             const float* coef, int32_t act, const float* gsums, double count, void* dy, void* stream);
/* stp3_conv2d_bn_bwd_apply_dx -- stp3_conv2d_bn_bwd_apply that ALSO writes the data gradient of the 1x1 convolution,
 * dx[m][ci] = sum_co dy[m][co] w[co][ci] (+ add[m][ci]: the gradient of the block's identity skip, may be NULL), [M][lddx] bf16:
 * the kernel holds the dy tile of its 32 pixels in LDS when it stores it, so the second matrix product reads it there instead
 * of a separate data-gradient convolution reading the 6x larger dy back from memory (the expand convolutions of the MBConv
 * blocks, stp3/models/encoder.py:57-97: 144 / 192 output channels, <= 32 input channels -- the whole-row streaming kernel;
 * anything else: STP3_EUNSUP, the caller keeps the two calls).  dy is still written (the weight gradient reads it). */
int stp3_conv2d_bn_bwd_apply_dx(const stp3_conv_dims* dims, const void* x, const void* w, const void* dz, int32_t ldz,
                                const float* coef, int32_t act, const float* gsums, double count, void* dy, void* dx,
                                int32_t lddx, const void* add, int32_t ldadd, void* stream);

/* stp3_conv2d_fwd_add -- stp3_conv2d_fwd (bf16 output, no statistics) with y = bf16(bf16(conv + bias) + add): the DATA GRADIENT
 * of a block's first convolution written together with the gradient of the block's skip connection (`add` = the gradient at
 * the block output, [N][Ho][Wo][ldadd >= Cout] bf16) -- what torch.autograd does with one more pass over both tensors when a
 * tensor has two consumers (the identity skips of the MBConv blocks, stp3/models/encoder.py:57-97, and of the decoder's
 * ResNet blocks, stp3/models/decoder.py:22-30).  Same bits as the separate addition (it adds the rounded convolution result).
 * Requires Cout % 8 == 0, ldy % 8 == 0, ldadd % 8 == 0 and 16-byte aligned y / add. */
int stp3_conv2d_fwd_add(const stp3_conv_dims* dims, const void* x, const void* w, const float* bias, const void* add,
                        int32_t ldadd, void* y, void* stream);

/* stp3_conv2d_wgrad -- dw[co][kh][kw][ci] = sum_{n,ho,wo} dy[n][ho][wo][co] * x[n][ho*stride-pad_h+kh*dil_h][..][ci]
 * (the weight gradient autograd derives for the convolutions above), float32 output in the weight's own
 * [Cout][KH][KW][Cin] layout.  dy [N][Ho][Wo][ldy >= Cout] and x [N][H][W][ldx >= Cin] are bf16; the pixel
 * contraction is split over workgroups and reduced deterministically through `workspace`
 * (stp3_conv2d_wgrad_workspace(dims) bytes).  Requires Cin % 4 == 0, Cout % 4 == 0, ldx % 4 == 0, ldy % 4 == 0.
 * dims.out_dtype / has_bias are ignored. */
int stp3_conv2d_wgrad_workspace(const stp3_conv_dims* dims, size_t* bytes);
int stp3_conv2d_wgrad(const stp3_conv_dims* dims, const void* dy, const void* x, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream);

/* The same weight gradient in two halves, for callers that can wait for it (the gradient of a LEAF parameter is read by
 * nobody before the optimizer -- torch.autograd's AccumulateGrad only keeps the tensor; reference: the `loss.backward()` /
 * `optimizer.step()` pair PyTorch-Lightning drives for stp3/trainer.py:101-172).
 * stp3_conv2d_wgrad_partials runs the split contraction only: partials[split][Cout][KH][KW][Cin] float32 into caller memory of
 * stp3_conv2d_wgrad_workspace(dims) bytes that must stay untouched until the reduction; *splits receives the split count.
 * stp3_conv2d_wgrad_reduce_batch sums the partials of n such layers in ONE launch per STP3_WGRAD_BATCH_MAX jobs (`jobs` is a
 * HOST array, it travels as the kernel argument): dw[i] = sum_k partials[k][i], the additions in the order of
 * stp3_conv2d_wgrad (bit-identical results). */
#define STP3_WGRAD_BATCH_MAX 96
typedef struct stp3_wgrad_job {
    const void* partials;   /* device: [splits][numel] float32 */
    float* dw;              /* device: [numel] float32 */
    int64_t numel;          /* Cout * KH * KW * Cin, < 2^32 */
    int32_t splits;
    int32_t reserved;
} stp3_wgrad_job;
int stp3_conv2d_wgrad_partials(const stp3_conv_dims* dims, const void* dy, const void* x, void* partials,
                               size_t partials_bytes, int32_t* splits, void* stream);
int stp3_conv2d_wgrad_reduce_batch(int32_t n, const stp3_wgrad_job* jobs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze-and-excitation data passes (csrc/stp3_se.hip).
 * Replace the pooling, the gate multiply and the elementwise / reduction passes of their backward inside the
 * EfficientNet MBConv blocks driven by stp3/models/encoder.py:57-97.
 *   x, dy, y : [N][rows][ld >= C] channels-last, dtype STP3_DTYPE_F32 / _BF16
 *   stp3_se_pool  : out [N][C] float32 = sum over rows of x (dy == NULL) or of dy * x (dy != NULL); deterministic
 *   stp3_se_scale : y = x * gate[n][c] + add[n][c]   (gate, add [N][C] float32; add may be NULL)
 */
typedef struct stp3_se_dims {
    int32_t N, rows, C, ld;
    int32_t dtype;
} stp3_se_dims;

int stp3_se_workspace_bytes(const stp3_se_dims* dims, size_t* bytes);
int stp3_se_pool(const stp3_se_dims* dims, const void* x, const void* dy, void* workspace, size_t workspace_bytes,
                 float* out, void* stream);
int stp3_se_scale(const stp3_se_dims* dims, const void* x, const float* gate, const float* add, void* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The two fully-connected layers of a squeeze-and-excitation block (csrc/stp3_se_mlp.hip): the MBConv gate
 *     gate = sigmoid(W2 swish(W1 mean_hw(x) + b1) + b2)
 * (efficientnet_pytorch MBConvBlock as driven by stp3/models/encoder.py:57-97) and its backward, float32.
 *   pooled_sum [N][C] : sum over the map (stp3_se_pool); the mean is pooled_sum * inv_rows
 *   w1 [S][C], b1 [S], w2 [C][S], b2 [C]
 *   fwd : z1 [N][S] (pre-activation, kept for the backward), gate [N][C]
 *   bwd : dgate [N][C] = sum_hw dy * x (stp3_se_pool with dy)  ->  dpooled [N][C] (gradient at pooled_sum... of the
 *         MEAN, already scaled by inv_rows, i.e. what stp3_se_scale adds per pixel), dw1 [S][C], db1 [S], dw2 [C][S],
 *         db2 [C]; dz2 [N][C] and dz1 [N][S] are caller-provided scratch.  Samples are reduced in ascending order.
 * Limits: S <= 8192; the weight-gradient kernel keeps 16 N + 8704 floats in LDS (N <= 480 samples per call); S * C < 2^31; else STP3_EUNSUP. */
typedef struct stp3_se_mlp_dims {
    int32_t N, C, S;
    float inv_rows;
} stp3_se_mlp_dims;

int stp3_se_mlp_fwd(const stp3_se_mlp_dims* dims, const float* pooled_sum, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* z1, float* gate, void* stream);
int stp3_se_mlp_bwd(const stp3_se_mlp_dims* dims, const float* dgate, const float* gate, const float* pooled_sum,
                    const float* z1, const float* w1, const float* w2, float* dz2, float* dz1, float* dpooled,
                    float* dw1, float* db1, float* dw2, float* db2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MBConv block passes (csrc/stp3_mbconv.hip, csrc/stp3_dwconv.hip): the EfficientNet block that
 * stp3/models/encoder.py:57-97 drives 22 times per image,
 *     E0 = expand(x) -> E1 = swish(BN0(E0)) -> E2 = depthwise(E1) -> S = swish(BN1(E2))
 *     -> gate = sigmoid(MLP(mean_hw S)) -> A = S * gate -> y = BN2(project(A)) (+ x),
 * with BatchNorm-1, its swish and the squeeze-excite gate applied where the data is consumed instead of in passes of
 * their own over the expanded tensor (the tensor `S` is never written), forward and backward.
 *
 *   stp3_dwconv2d_fwd_stats : stp3_dwconv2d_fwd + sums [2][C] = per-channel sum / sum of squares of the ROUNDED outputs
 *                             (the statistics of BN1; no stp3_bn_stats pass over E2).  workspace:
 *                             stp3_dwconv2d_fwd_stats_workspace(dims) bytes; deterministic.
 *   stp3_bn_finalize        : sums [2][C] over `count` elements -> coef [4][C] = scale (= gamma * invstd), shift
 *                             (= beta - mean * scale), mean, invstd; updates running_mean / running_var (momentum, unbiased
 *                             variance) when non-NULL -- the arithmetic of stp3_bn_apply_fwd's training mode, bit for bit.
 *   stp3_se_pool_act        : out [N][C] = sum over rows of act(scale[c] * x + shift[c])    (the squeeze of S from E2)
 *   stp3_mbconv_scale_act   : y = act(scale[c] * x + shift[c]) * gate[n][c]                  (A from E2; gate may be NULL)
 *   stp3_mbconv_bwd_reduce  : ONE pass over (da = dL/dA, x = E2): sums5 [5][N][C] (quantity-major) =
 *                               [0] sum da*S  (the gate's gradient, input of stp3_se_mlp_bwd)
 *                               [1] sum da*S'   [2] sum da*S'*xhat   [3] sum S'   [4] sum S'*xhat
 *                             with S = act(pre), S' = act'(pre), pre = scale*x + shift, xhat = (x - mean) * invstd.
 *   stp3_mbconv_bwd_coef    : gsums [2][C]:  sum g = sum_n gate*[1] + dpooled*[3],  sum g*xhat = sum_n gate*[2] + dpooled*[4]
 *                             where g = (da*gate + dpooled) * S' is the gradient at the BatchNorm-1 output
 *                             (dbeta1 = gsums[0], dgamma1 = gsums[1]); the host may all-reduce gsums over ranks.
 *   stp3_mbconv_bwd_apply   : dx = scale * (g - gsums[0]/count - xhat * gsums[1]/count)   (= dL/dE2), same dtype as x
 *   x, da, dx, y : [N][rows][ld] channels-last, dims.dtype (bf16: C, ld multiples of 8; f32: of 4; 16-byte aligned),
 *   dims.ld = row stride of x; ldg / ldy = row stride of da and dx / of y.  coef as written by stp3_bn_finalize.
 *   gate, dpooled [N][C] float32 (dpooled: the per-pixel term stp3_se_mlp_bwd returns).  workspace:
 *   stp3_mbconv_workspace_bytes(dims).  Deterministic, no atomics.
 */
int stp3_dwconv2d_fwd_stats_workspace(const stp3_dwconv_dims* dims, size_t* bytes);
int stp3_dwconv2d_fwd_stats(const stp3_dwconv_dims* dims, const void* x, const float* w, void* y, float* sums,
                            void* workspace, size_t workspace_bytes, void* stream);
int stp3_bn_finalize(const float* sums, int32_t C, double count, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* coef, void* stream);
int stp3_mbconv_workspace_bytes(const stp3_se_dims* dims, size_t* bytes);
int stp3_se_pool_act(const stp3_se_dims* dims, const void* x, const float* scale, const float* shift, int32_t act,
                     void* workspace, size_t workspace_bytes, float* out, void* stream);
int stp3_mbconv_scale_act(const stp3_se_dims* dims, int32_t ldy, const void* x, const float* scale, const float* shift,
                          int32_t act, const float* gate, void* y, void* stream);
int stp3_mbconv_bwd_reduce(const stp3_se_dims* dims, int32_t ldg, const void* da, const void* x, const float* coef,
                           int32_t act, void* workspace, size_t workspace_bytes, float* sums5, void* stream);
int stp3_mbconv_bwd_coef(int32_t N, int32_t C, const float* sums5, const float* gate, const float* dpooled,
                         float* gsums, void* stream);
int stp3_mbconv_bwd_apply(const stp3_se_dims* dims, int32_t ldg, const void* da, const void* x, const float* coef,
                          int32_t act, const float* gate, const float* dpooled, const float* gsums, double count,
                          void* dx, void* stream);

/* stp3_dwconv2d_fwd_stats_bn -- (depthwise convolution -> BatchNorm-1 of an MBConv block, stp3/models/encoder.py:57-97)
 * stp3_dwconv2d_fwd_stats + stp3_bn_finalize in the launches of the former (one process: nothing
 * happens between the statistics and their use, so the final reduction finishes its own channels -- coef [4][C] = scale |
 * shift | mean | invstd, running statistics updated; the additions keep the order of the stand-alone reduction and the
 * constants stp3_bn_finalize's arithmetic: bit-identical).  count = elements per channel. */
int stp3_dwconv2d_fwd_stats_bn(const stp3_dwconv_dims* dims, const void* x, const float* w, void* y, float* sums, double count,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* coef, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training losses of the perception path and the label warp (csrc/stp3_loss.hip).
 *
 * stp3_ce_topk_fwd / _bwd -- weighted cross-entropy with ignore_index, per-row scale (future discount) and the mean of
 * the k largest per-pixel losses of every row: stp3/losses.py:43-83 (SegmentationLoss: rows = B*S), :85-114 (HDmapLoss,
 * one call per map element: rows = B), :116-134 (DepthLoss: k = 0, no selection).
 *   logits  element (row, c, p) at logits[row*stride_row + c*stride_c + p*stride_p], dims.dtype f32 / bf16
 *   labels  [rows][P] int64;  class_weights [C] float32 or NULL;  row_scale [rows] float32 or NULL
 *   loss_px [rows][P] float32 out: scale[row] * w[y] * (logsumexp(z) - z[y]), 0 for ignored pixels (kept for the backward)
 *   sel     [rows][2] float32 out: tau = the k-th largest loss of the row (found by a radix select on the float bits; the
 *           reference sorts the row, losses.py:76-81) and the share of the elements EQUAL to tau in the top k
 *   out[0]  = (accumulate ? out[0] : 0) + out_scale * sum_rows sum_{k largest} loss      (out_scale = weight / (rows*k))
 *   bwd     dlogits (same strides / dtype as logits) = gout[0] * out_scale * take * scale[row] * w[y] * (softmax - onehot),
 *           take = 1 above tau, the tie share at tau, 0 below: ties at the threshold share their part of the gradient
 *           equally (torch.topk picks arbitrary ones; the loss value and the total gradient mass are the same)
 *   k <= 0 or k >= P: no selection (mean over all pixels: out_scale = weight / (rows*P)).
 * stp3_reg_loss_fwd / _bwd -- stp3/losses.py:6-40 SpatialRegressionLoss: pred, target [rows][C][P] contiguous;
 *   out[0] = sum over pixels with target[row][0][p] != ignore_value of scale[row] * sum_c |d| (norm 1) or d^2 (norm 2),
 *   divided by the number of such pixels (0 when there are none); out[1] = that number.  bwd: stat = out of the forward.
 * stp3_warp_nearest -- stp3/utils/geometry.py:196-238 warp_features with mode='nearest' on label maps
 *   (F.affine_grid + F.grid_sample, zeros padding, align_corners=False): x, y [frames][C][H][W] float32, theta [frames][6]
 *   (row-major 2x3), identity [frames] int32 or NULL (non-zero: the frame is copied unchanged).
 * Deterministic; float32 arithmetic; workspaces: stp3_ce_topk_workspace_bytes / stp3_reg_loss_workspace_bytes.
 */
typedef struct stp3_ce_dims {
    int32_t rows, P, C, k;
    int32_t ignore_index, dtype;
    int64_t stride_row, stride_c, stride_p;
} stp3_ce_dims;

int stp3_ce_topk_workspace_bytes(const stp3_ce_dims* dims, size_t* bytes);
int stp3_ce_topk_fwd(const stp3_ce_dims* dims, const void* logits, const int64_t* labels, const float* class_weights,
                     const float* row_scale, float* loss_px, float* sel, double out_scale, int32_t accumulate, float* out,
                     void* workspace, size_t workspace_bytes, void* stream);
int stp3_ce_topk_bwd(const stp3_ce_dims* dims, const void* logits, const int64_t* labels, const float* class_weights,
                     const float* row_scale, const float* loss_px, const float* sel, const float* gout, double out_scale,
                     void* dlogits, void* stream);
int stp3_reg_loss_workspace_bytes(size_t* bytes);
int stp3_reg_loss_fwd(int32_t rows, int32_t C, int32_t P, int32_t norm, float ignore_value, int32_t dtype, const void* pred,
                      const float* target, const float* row_scale, float* out, void* workspace, size_t workspace_bytes,
                      void* stream);
int stp3_reg_loss_bwd(int32_t rows, int32_t C, int32_t P, int32_t norm, float ignore_value, int32_t dtype, const void* pred,
                      const float* target, const float* row_scale, const float* stat, const float* gout, void* dpred,
                      void* stream);
int stp3_warp_nearest(int32_t frames, int32_t C, int32_t H, int32_t W, const float* x, const float* theta,
                      const int32_t* identity, float* y, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 shadow copies of all convolution weights in ONE launch (csrc/stp3_wprep.hip).  Replaces the per-layer cast / flip / transpose / re-layout the host would
 * otherwise redo after every optimizer step for the operands of stp3_conv2d_fwd (forward: [Cout][KH][KW][Cin];
 * data gradient: [Cin][KH][KW][Cout] with the taps flipped).
 *   table : n_entries structs in DEVICE memory, sorted by first_block
 *   src   : fp32 master weight, element (co, ci, r, s) at src[co*stride_co + ci*stride_ci + r*stride_kh + s*stride_kw]
 *   fwd / flip : bf16 destinations (either may be NULL)
 *   first_block : exclusive scan over the table of ceil(cout*cin*kh*kw / 256); total_blocks = the scan's total
 * Rounding: to nearest even, as torch's .to(bfloat16). */
typedef struct stp3_wprep_entry {
    const float* src;
    void* fwd;
    void* flip;
    int64_t stride_co, stride_ci, stride_kh, stride_kw;
    int64_t first_block;
    int32_t cout, cin, kh, kw;
    /* a PIECE of an assembled weight: the block lands at channel offsets (co_off, ci_off) of destinations with dst_cout x
     * dst_cin channels (fwd [dst_cout][KH][KW][dst_cin], flip [dst_cin][KH][KW][dst_cout]); dst_cout == 0: the entry is the
     * whole weight (dst = cout x cin, offsets 0).  Elements no piece covers are not written. */
    int32_t dst_cout, dst_cin, co_off, ci_off;
    int32_t fwd_f32;       /* 1: fwd receives float32 (the tap-major [KH*KW][C] weights of stp3_dwconv2d_*: cout = 1, cin = C) */
    int32_t reserved;
} stp3_wprep_entry;

int stp3_conv2d_prep_weights(const stp3_wprep_entry* table, int32_t n_entries, int64_t total_blocks, void* stream);

/* The way back: the float32 gradient of an assembled weight ([dst_cout][KH][KW][dst_cin], what stp3_conv2d_wgrad writes) cut
 * into its pieces, each stored into its parameter's gradient -- same table layout, with ``src`` = the (writable) destination
 * inside the parameter's gradient, read through the entry's strides, and ``fwd`` = the assembled gradient.  One launch for
 * every assembled weight of a backward pass.  Replaces the backward of the reference's weight plumbing (torch autograd of
 * the slices / pads / concatenations around stp3/layers/temporal.py:8-37 CausalConv3d, stp3/layers/convolutions.py ASPP.project,
 * stp3/models/decoder.py:96-140 heads): slice_backward, constant_pad_nd, cat and add kernels per layer. */
int stp3_conv2d_scatter_weight_grads(const stp3_wprep_entry* table, int32_t n_entries, int64_t total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient-norm clipping + Adam on flat fp32 buckets in three launches (csrc/stp3_optim.hip).  Replaces, for the flat buckets of stp3_amd/parallel.py, the reference's
 * gradient_clip_val (train.py:48, torch clip_grad_norm_ semantics: scale = min(max_norm / (norm + 1e-6), 1)) and
 * torch.optim.Adam step (trainer.py:456-462: L2 weight decay folded into the gradient, bias correction, no amsgrad).
 *   table : n_buckets structs in DEVICE memory, sorted by first_block; grad / param / exp_avg / exp_avg_sq are
 *           float32 arrays of numel elements; first_block = exclusive scan of ceil(numel / 4096)
 *   state : 5 floats in device memory, [0] = step count (in/out, incremented by the call), out: [1] clip scale,
 *           [2] lr / (1 - beta1^t), [3] sqrt(1 - beta2^t), [4] total gradient norm before clipping
 *   max_norm <= 0 : no clipping.  The clipped gradient is written back to grad (as clip_grad_norm_ does).
 *   workspace : stp3_optim_workspace_bytes(total_blocks) bytes.  No host synchronisation; deterministic. */
typedef struct stp3_optim_bucket {
    float* grad;
    float* param;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    int64_t first_block;
} stp3_optim_bucket;

int stp3_optim_workspace_bytes(int64_t total_blocks, size_t* bytes);
int stp3_optim_clip_adam(const stp3_optim_bucket* table, int32_t n_buckets, int64_t total_blocks, float max_norm,
                         float lr, float beta1, float beta2, float eps, float weight_decay, float* state,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Causal two-frame channel pairing of the frame-folded BEV sequence (csrc/stp3_temporal.hip): the operand of the
 * (2,3,3) convolution of CausalConv3d, stp3/layers/temporal.py:252-273 (time padded on the left by one frame:
 * y[t] = W[:, :, 0] * x[t-1] + W[:, :, 1] * x[t], x[-1] = 0), as ONE 2-D convolution over 2C channels.
 *   x  [frames][rows][ldx >= C]  bf16 / float32, channels-last; frames = B*T, the T frames of a sample consecutive
 *   y  [frames][rows][2C] : y[n][r][0:C] = x[n-1][r][0:C] (zero when n % T == 0) ; y[n][r][C:2C] = x[n][r][0:C]
 *   bwd: dy [frames][rows][2C] -> dx [frames][rows][C] (dense),
 *        dx[n][r][c] = dy[n][r][C + c] + (dy[n+1][r][c] unless n % T == T - 1)      (float32 add, rounded once)
 * C and ldx multiples of 16 bytes (8 bf16 / 4 float32), 16-byte aligned pointers (STP3_EUNSUP otherwise). */
typedef struct stp3_pair_dims {
    int32_t frames, T, rows, C, ldx, dtype;
} stp3_pair_dims;
int stp3_causal_pair_fwd(const stp3_pair_dims* dims, const void* x, void* y, void* stream);
int stp3_causal_pair_bwd(const stp3_pair_dims* dims, const void* dy, void* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bilinear up-sampling by an integer factor, align_corners = False (csrc/stp3_upsample.hip): nn.Upsample(scale_factor=2,
 * mode='bilinear') of UpsamplingConcat / UpsamplingAdd, stp3/layers/convolutions.py:183-215 (encoder merge and the three
 * BEV decoder stages).
 *   x  [N][H][W][ldx >= C]            bf16 / float32, channels-last (ldx: a channel slice is read in place)
 *   y  [N][H*scale][W*scale][ldy >= C] same type (ldy: the result may be written into a slice of a concatenation)
 *   y[oh][ow] = l0h * (l0w * x[h0][w0] + l1w * x[h0][w1]) + l1h * (l0w * x[h1][w0] + l1w * x[h1][w1]),
 *       src = max((o + 0.5) / scale - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, size - 1), l1 = src - i0, l0 = 1 - l1
 *       (float32 arithmetic in this order -- torch's -- then one rounding to the element type)
 *   bwd: dy [N][H*scale][W*scale][ldy] -> dx [N][H][W][ldx]: every input pixel gathers the gradients of the outputs
 *        that read it (float32 accumulation, deterministic, no atomics)
 * scale 1..4; C, ldx, ldy multiples of 16 bytes; 16-byte aligned pointers; < 2^31 vectors (STP3_EUNSUP otherwise). */
typedef struct stp3_upsample_dims {
    int32_t N, H, W, C, scale, ldx, ldy, dtype;
} stp3_upsample_dims;
int stp3_upsample_bilinear_fwd(const stp3_upsample_dims* dims, const void* x, void* y, void* stream);
int stp3_upsample_bilinear_bwd(const stp3_upsample_dims* dims, const void* dy, void* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Planner: trajectory-cost evaluation (csrc/stp3_plan.hip) -- the reference's Cost_Function.forward, stp3/cost.py:26-47,
 * with its seven terms (SafetyCost :210-241, HeadwayCost :244-272, LR_divider :274-315, Comfort :318-372, Progress
 * :374-392, Rule :183-207, Cost_Volume :166-181) in one launch.  Called by Planning.select / Planning.loss
 * (stp3/models/planning_model.py:43-88) once for the N sampled trajectories and once for the expert trajectory (N = 1).
 *   trajs        [B][N][T][2] float32   (lateral, forward) metres in the ego frame, UNflipped (the kernel applies the
 *                                       reference's (-1, 1) flip, cost.py:35)
 *   cost_volume  [B][T][H][W] float32   the decoder's cost-volume head, raw (clamped to [0, 1000] inside)
 *   occupancy    [B][T][H][W] float32   0 / 1 (labels in training, argmax of the prediction in evaluation)
 *   drivable     [B][H][W]    float32   drivable-area mask AFTER the reference's preprocessing (labels as they are;
 *                                       logits: softmax, entries < 0.5 zeroed -- cost.py:196-201; host side, three operators)
 *   lane         [B][H][W]    float32   lane-divider mask after the same preprocessing (:289-294); != 0 is a divider
 *   target       [B][2] float32 ; target_sum [1] float32 = sum of `target` (progress drops its goal term when the sum
 *                                       over the batch is < 0.5, :386 -- a device scalar, no host synchronisation)
 *   footprint0 / footprint_lambda  [K0][2] / [KL][2] int32  (row, column) cells of the ego box and of the box inflated
 *                                       by LAMBDA, as BaseCost.get_origin_points builds them (:70-83; 32 / 192 cells)
 *   cost_fc [B][N] float32 = comfort + progress ; cost_fo [B][N][T] float32 = safety + headway + lane + volume + rule,
 *                                       every term clamped as in :36-42
 *   cv_cell [B][N][T] int32 / cv_scale [B][N][T] float32: (optional, both or neither) the cost-volume cell each cost
 *                                       read and d cost_fo / d cost_volume[cell] -- what the backward needs
 *   bwd: grad_cost_volume [B][T][H][W] float32, written completely (zeros where nothing was read); contributions of
 *        several trajectories to one cell are added in ascending trajectory order (deterministic, no atomics).
 * Limits: B*N*T and B*T*H*W < 2^31; backward N <= 20 480 (STP3_EUNSUP). */
typedef struct stp3_plan_dims {
    int32_t B, N, T, H, W, K0, KL;
    float dx0, dx1, bx0, bx1;                     /* BEV resolution / first cell centre, (x = forward, y = side) */
    float safety, headway, lrdivider, comfort, progress, volume, rule;   /* COST_FUNCTION factors; rule = 5 */
    float w0, w1;                                 /* SafetyCost.w */
    float headway_dist, lr_dist;                  /* 10 m, 1 m */
} stp3_plan_dims;
int stp3_traj_cost_fwd(const stp3_plan_dims* dims, const float* trajs, const float* cost_volume, const float* occupancy,
                       const float* drivable, const float* lane, const float* target, const float* target_sum,
                       const int32_t* footprint0, const int32_t* footprint_lambda, float* cost_fc, float* cost_fo,
                       int32_t* cv_cell, float* cv_scale, void* stream);
int stp3_traj_cost_bwd(const stp3_plan_dims* dims, const float* grad_cost_fo, const int32_t* cv_cell,
                       const float* cv_scale, float* grad_cost_volume, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Camera images: decoded bytes -> network input (csrc/stp3_image.hip).  The per-image chain of the reference's loader,
 * stp3/datas/NuscenesData.py:236-244: resize_and_crop_image (stp3/utils/geometry.py:9-13: PIL resize BILINEAR + crop)
 * followed by torchvision ToTensor + Normalize (:68-72), for all N images of a batch in one launch.
 *   images   [N][H][W][3] uint8 (RGB, as PIL decodes them)
 *   kk_h / bounds_h  [Wr][ksize_h] int32 / [Wr][2] int32: Pillow's horizontal resampling coefficients in its 22-bit fixed
 *            point and (first source column, count) per resized column; kk_v / bounds_v the same for the rows.  Built on
 *            the host exactly as Pillow builds them (stp3_amd.datas.pil_bilinear_coefficients); configuration-only.
 *   out      [N][3][Ho][Wo] float32 or bf16 = ((byte / 255) - mean[c]) / std[c], byte = the resized image at
 *            (top + r, left + x); window positions outside the resized image are PIL's zero padding.
 *   strip_rows: the largest number of source rows the vertical taps of the output rows of one workgroup span (host:
 *            from bounds_v; stp3_image_prep_rows_per_workgroup() returns how many rows that is); sizes the LDS strip.
 * The resized bytes equal PIL.Image.resize(..., BILINEAR) bit for bit (both passes, byte rounding between them). */
typedef struct stp3_image_dims {
    int32_t N, H, W;              /* source */
    int32_t Wr, Hr;               /* resize_dims */
    int32_t left, top, Wo, Ho;    /* crop window in the resized image */
    int32_t ksize_h, ksize_v;
    int32_t out_dtype;            /* STP3_DTYPE_F32 | STP3_DTYPE_BF16 */
    float mean[3], std[3];
} stp3_image_dims;
int stp3_image_prep_rows_per_workgroup(void);
int stp3_image_prep_lds_bytes(const stp3_image_dims* dims, int32_t strip_rows, size_t* bytes);
int stp3_image_prep(const stp3_image_dims* dims, const uint8_t* images, const int32_t* kk_h, const int32_t* bounds_h,
                    const int32_t* kk_v, const int32_t* bounds_v, int32_t strip_rows, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone voxel summing (csrc/stp3_voxsum.hip): the operator-level twin of the reference's
 * VoxelsSumming.forward / .backward (stp3/utils/geometry.py:302-318 / :320-330) for callers that hold the
 * rank-sorted row matrix; the fused path above never builds it.
 *   x        [M][channels] float32, rows sorted by voxel rank
 *   seg_off  [n_segments + 1] int32, ascending, seg_off[0] = 0, seg_off[n_segments] = M: voxel s owns the rows
 *            [seg_off[s], seg_off[s+1])  (the host derives it from ranks[1:] != ranks[:-1], geometry.py:308-309)
 *   fwd: out [n_segments][channels] = per-voxel row sums, rows added in ascending order (deterministic; no
 *        running sum over the whole matrix, so no cancellation against preceding voxels)
 *   bwd: grad_x [M][channels], every row receives its voxel's grad_out row (geometry.py:326-328)
 * n_segments == 0 is a no-op. */
int stp3_voxels_sum_fwd(const float* x, const int32_t* seg_off, int32_t n_segments, int32_t channels, float* out,
                        void* stream);
int stp3_voxels_sum_bwd(const float* grad_out, const int32_t* seg_off, int32_t n_segments, int32_t channels,
                        float* grad_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BEV labels of the data loader (csrc/stp3_labels.hip).
 *
 * stp3_fill_polygons -- cv2.fillPoly of integer polygons, in order, a later one overwriting an earlier one: what
 *   stp3/datas/NuscenesData.py:303-338 (get_birds_eye_view_label: instance / segmentation / pedestrian maps from the
 *   annotation boxes) and :520-564 (road polygons of voxelize_hd_map) call.  OpenCV is a third-party dependency that is
 *   not vendored under the reference: restated from its published algorithm (8-connected Bresenham outline drawn left to
 *   right + scanlines between pairs of active edges, ceil(left) .. floor(right), 16.16 fixed-point edge positions with the
 *   slope truncated), PARITY UNPINNED; cross-checked against an edge-walking restatement and against Pillow.
 *   polys [n_poly] in paint order: `map` = which of the n_maps [H][W] float32 images, nv <= 8 vertices xy = (column, row)
 *   as cv2 takes them, `value` painted.  The maps are painted over what they hold (zero them for fresh labels).
 * stp3_instance_labels -- stp3/utils/instance.py:12-77 convert_instance_mask_to_center_and_offset_label:
 *   instance [T][H][W] int64 ids (0 = background, 1..K instances); warped [T][H][W] float32 = frame t's instance map warped
 *   into frame t-1 (warp_features(..., mode='nearest'), frame 0 unused) or NULL (no displacement labels);
 *   center [T][1][H][W] = max over the frame's instances of exp(-((xc - x)^2 + (yc - y)^2) / sigma^2), (xc, yc) = rounded
 *   mean pixel of the instance; offset [T][2][H][W] = (xc - x, yc - y) on the instance's pixels, ignore_index elsewhere;
 *   flow [T][2][H][W] = (rounded mean pixel of the instance's WARPED next-frame mask) - (xc, yc) on the instance's
 *   pixels when the instance is in the next frame too and its warped mask is not empty, ignore_index elsewhere.
 *   workspace: stp3_instance_labels_workspace_bytes(T, K) (integer moments; zeroed by the call).  Deterministic.
 */
typedef struct stp3_poly {
    int32_t map, nv;
    float value;
    int32_t reserved;
    int32_t xy[16];
} stp3_poly;

int stp3_fill_polygons(const stp3_poly* polys, int32_t n_poly, int32_t n_maps, int32_t H, int32_t W, float* maps,
                       void* stream);
int stp3_instance_labels_workspace_bytes(int32_t T, int32_t K, size_t* bytes);
int stp3_instance_labels(int32_t T, int32_t H, int32_t W, int32_t K, float ignore_index, float sigma,
                         const int64_t* instance, const float* warped, void* workspace, size_t workspace_bytes,
                         float* center, float* offset, float* flow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STP3_HIP_H */
