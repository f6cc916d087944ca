/*
 * tripled_hip.h -- C ABI of libtripled_hip.so: hand-written CDNA4 (gfx950) kernels for
 * the self-supervised depth loss hot path.
 *
 * The reference (ufukpage/TripleD, pure Python/PyTorch) has NO native interface on this
 * path; its hot ops are nn.Module methods built from stock ATen calls.  Each entry point
 * below names the reference code it replaces (paths relative to the reference checkout).
 * The binding a maintainer of the reference would add is a ctypes stub: INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter comment says "host";
 *   - tensors are dense, NCHW, fp32, row-major (x fastest), exactly as the reference holds them -- except the colour
 *     frames of td_photo_fwd, which are RGBX pixels [B,H,W,4] produced from the reference's [B,3,H,W] tensors by
 *     td_photo_identity (as a by-product) or td_pack_rgbx (one 16-byte load per pixel or bilinear tap: the forward is bound
 *     by the number of memory instructions in flight, not by bytes); td_photo_identity and td_photo_bwd read the NCHW frames;
 *   - the caller owns every buffer; the library allocates nothing and keeps no global state;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream); no call synchronises, so every call is legal inside hipGraph capture;
 *   - return value: TD_OK (0) or a negative TD_ERR_* code; nothing throws across the ABI;
 *   - n_src = number of non-reference frames (frame_ids[1:]), 1..TD_MAX_SRC;
 *   - candidate order of the min-reprojection is the reference's: identity terms for
 *     src 0..n_src-1 first (when automasking), then the warped terms for src 0..n_src-1.
 */
#ifndef TRIPLED_HIP_H
#define TRIPLED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TD_ABI_VERSION 3      /* 3: fused bottleneck entry points (td_conv1x1_fwd_bnrelu, td_conv1x1_dgrad*, td_bn_*_partials); 2: td_photo_fwd takes RGBX frames (td_photo_identity emits them), idloss is [B,H,W,n_src], d_up has one plane per frame */
#define TD_MAX_SRC 4

#define TD_OK 0
#define TD_ERR_BAD_ARG (-1)      /* null pointer / non-positive size / n_src out of range */
#define TD_ERR_UNSUPPORTED (-2)  /* shape the kernels do not cover (e.g. H or W < 3)        */
#define TD_ERR_LAUNCH (-3)       /* hipGetLastError() != hipSuccess after the launch        */
#define TD_ERR_WORKSPACE (-4)    /* caller-provided workspace too small                     */

typedef void* td_stream_t;

int td_abi_version(void);
const char* td_error_string(int code);
/* Last HIP error text recorded by a failed launch on this thread ("" if none). */
const char* td_last_hip_error(void);

/* Number of thread blocks td_photo_fwd uses for a B x H x W problem (sizes `partial`), and
 * the number td_photo_bwd uses (sizes `dP_partial`; its tiles are smaller). */
int td_photo_num_blocks(int B, int H, int W);
int td_photo_bwd_num_blocks(int B, int H, int W);

/* [B,3,H,W] fp32 colour frame -> RGBX [B,H,W,4] (x = 0), the frame format of td_photo_fwd (when td_photo_identity does not run). */
int td_pack_rgbx(const float* img, int B, int H, int W, float* out, td_stream_t stream);

/*
 * Identity (auto-mask) photometric term, scale independent:
 *   idloss[b, i, y, x] = 0.85 * mean_c SSIM(src_i, tgt) + 0.15 * mean_c sqrt((tgt - src_i)^2 + 1e-6)
 * Replaces compute_reprojection_loss(inputs[("color", f, 0)], target) inside the automask loop,
 * mono/model/mono_fm_joint_inpaint/net.py:101-106 (SSIM: mono/model/mono_fm_joint/layers.py:85-107,
 * robust_l1 + weights: mono/model/mono_fm_joint/net.py:59-71).  The reference recomputes it per
 * scale; it does not depend on the scale, so it is computed once per step here.
 *   tgt     [B,3,H,W] (the reference's tensor)
 *   src     host array of n_src device pointers, each [B,3,H,W]
 *   idloss  [B,H,W,n_src] (out): the terms of a pixel are adjacent (td_photo_fwd reads them with one load)
 *   tgt_rgbx, src_rgbx (out, both or neither NULL): [B,H,W,4] RGBX copies of the frames (x = 0), the frame format of
 *           td_photo_fwd -- written from the pixels this kernel reads anyway, which saves the td_pack_rgbx passes
 */
int td_photo_identity(const float* tgt, const float* const* src, int n_src,
                      int B, int H, int W, float* idloss, float* tgt_rgbx, float* const* src_rgbx, td_stream_t stream);

/*
 * Fused per-scale photometric forward.  Replaces, for one scale,
 *   generate_images_pred            mono/model/mono_fm_joint/net.py:181-194
 *     F.interpolate(disp,[H,W],bilinear)            :183
 *     disp_to_depth                                 :157-162
 *     Backproject.forward                           mono/model/mono_fm_joint/layers.py:57-61
 *     Project.forward                               mono/model/mono_fm_joint/layers.py:73-82
 *     F.grid_sample(img, grid, padding_mode="border")  net.py:193  (bilinear, align_corners=False)
 *   compute_reprojection_loss (SSIM + robust L1)    mono/model/mono_fm_joint/net.py:67-71
 *   the automask + torch.cat + torch.min block      mono/model/mono_fm_joint_inpaint/net.py:101-117
 *
 *   tgt, src  RGBX frames [B,H,W,4] (from td_photo_identity's copies or td_pack_rgbx)
 *   disp      [B,1,hs,ws]  sigmoid disparity of this scale (any hs<=H, ws<=W)
 *   P         [n_src,B,3,4]  (K @ T_i)[:, :3, :]  -- formed by the caller (tiny matmul, keeps autograd to T)
 *   invK      [B,4,4]  (only the upper-left 3x3 block is read, as in the reference)
 *   idloss    [B,H,W,n_src] from td_photo_identity, or NULL to disable automasking
 *   noise     [n_src,B,H,W] standard-normal draws (scaled by 1e-5 inside, net.py:105) or NULL
 *   argmin    [B,H,W] uint8 (out): index into the candidate list (reference: int64 "min_index")
 *   warped    [n_src,B,3,H,W] (out, nullable): the warped sources, outputs[("color", f, s)]
 *   min_map   [B,H,W] (out, nullable): per-pixel minimum
 *   partial   [td_photo_num_blocks] (out): per-wave-task sums of the per-pixel minimum; the loss is
 *             sum(partial) / (B*H*W) / n_scales  (td_sum_scaled finishes it deterministically)
 *   coef      [B,9,H,W] (out, nullable; pass it when a backward will follow): per pixel p and channel c
 *             the SSIM-adjoint coefficients (alpha, beta, gamma) of the warped frame the arg-min selected
 *             (zeros where an identity term won), with  d SSIM_p / d x_q = (alpha + beta x_q + gamma y_q)/9
 *             for every member q of p's 3x3 window.
 */
int td_photo_fwd(const float* tgt, const float* const* src, int n_src,
                 const float* disp, const float* P, const float* invK,
                 const float* idloss, const float* noise,
                 int B, int H, int W, int hs, int ws,
                 float min_depth, float max_depth,
                 uint8_t* argmin, float* warped, float* min_map, float* partial, float* coef,
                 td_stream_t stream);

/*
 * Backward of td_photo_fwd w.r.t. disp and P (what autograd derives in the reference from the
 * same lines).  The warp is recomputed in-kernel; argmin and coef come from the forward.
 *   gscale    device scalar: d(total)/d(loss_s) as handed over by autograd
 *   inv_count 1 / (B*H*W*n_scales)  (the mean and the /len(scales) of net.py:117)
 *   d_up      [n_src,B,H,W] (out, workspace): gradient w.r.t. the UPSAMPLED disparity, one plane per source frame
 *             (the frames of a column strip are separate wave tasks; their sum is the gradient)
 *   dP_partial[td_photo_bwd_num_blocks, n_src*12] (out): per-block partial sums of dL/dP
 *   tgt, src  the NCHW frames [B,3,H,W]; tgt_rgbx, src_rgbx (both or neither NULL): their RGBX copies -- read instead of the
 *             NCHW frames at the finest scale (2*hs >= H), where the disparity's detail breaks the coalescing of dword gathers
 * Follow with td_upsample_adjoint_planes (d_up planes -> d_disp) and td_reduce_dP.
 */
int td_photo_bwd(const float* tgt, const float* const* src, const float* tgt_rgbx, const float* const* src_rgbx, int n_src,
                 const float* disp, const float* P, const float* invK,
                 const uint8_t* argmin, const float* coef, int automask,
                 const float* gscale, float inv_count,
                 int B, int H, int W, int hs, int ws,
                 float min_depth, float max_depth,
                 float* d_up, float* dP_partial, td_stream_t stream);

/* Adjoint of F.interpolate(..., [H,W], mode="bilinear", align_corners=False)
 * (mono/model/mono_fm_joint/net.py:183): d_up [B,H,W] -> d_disp [B,1,hs,ws].
 * accumulate != 0 adds into d_disp instead of overwriting it. Gather form, deterministic. */
int td_upsample_adjoint(const float* d_up, int B, int H, int W, int hs, int ws,
                        float* d_disp, int accumulate, td_stream_t stream);
/* The same adjoint of the SUM of n_planes gradient planes d_up [n_planes,B,H,W] (td_photo_bwd's per-frame planes), summed in
 * plane order while gathering. */
int td_upsample_adjoint_planes(const float* d_up, int n_planes, int B, int H, int W, int hs, int ws,
                               float* d_disp, int accumulate, td_stream_t stream);

/* dP[i,b,:,:] = sum over the blocks of sample b of dP_partial (deterministic tree). */
int td_reduce_dP(const float* dP_partial, int n_src, int B, int H, int W,
                 float* dP /* [n_src,B,3,4] */, td_stream_t stream);

/* out[0] = scale * sum(partial[0..n)) ; deterministic single-block tree. */
int td_sum_scaled(const float* partial, int n, float scale, float* out, td_stream_t stream);

/*
 * Area (box) down-sampling of the target frame: F.interpolate(img, (h, w), mode='area')
 * at mono/model/mono_fm_joint/net.py:283 and :311, for integer factors fy = H/h, fx = W/w.
 *   img [B,C,H,W] -> out [B,C,h,w]
 */
int td_area_downsample(const float* img, int B, int C, int H, int W, int h, int w,
                       float* out, td_stream_t stream);

/* Number of blocks td_smooth_fwd/bwd use for a B x h x w disparity. */
int td_smooth_num_blocks(int B, int h, int w);

/*
 * Edge-aware first + second order smoothness, forward.  Replaces
 *   disp mean-normalisation   mono/model/mono_fm_joint_inpaint/net.py:122-124
 *   get_smooth_loss + gradient  mono/model/mono_fm_joint/net.py:279-307
 * (the image is the area-resized target, see td_area_downsample).
 *   disp    [B,1,h,w]   img [B,3,h,w]
 *   normalize != 0 applies disp / (mean_hw(disp) + 1e-7) per sample first
 *   mean    [B] (out): per-sample mean of disp (saved for the backward)
 *   partial [td_smooth_num_blocks, 6] (out): per-block sums of the six terms
 *           (dx, dy, dxx, dxy, dyx, dyy); each term's mean uses its own element count.
 * Finish with td_smooth_finish.
 */
int td_smooth_fwd(const float* disp, const float* img, int B, int h, int w, int normalize,
                  float* mean, float* partial, td_stream_t stream);

/* loss[0] = weight * sum_k (sum_blocks partial[:,k]) / count_k   (count_k = elements of term k). */
int td_smooth_finish(const float* partial, int B, int h, int w, float weight, float* loss,
                     td_stream_t stream);

/*
 * Smoothness backward: d(loss)/d(disp) including the mean-normalisation.
 *   gscale device scalar upstream gradient; weight as in td_smooth_finish
 *   g_hat  [B,h,w] (workspace): gradient w.r.t. the normalised disparity
 *   dot_partial [td_smooth_num_blocks] (workspace): per-block sums of g_hat * disp
 *   d_disp [B,1,h,w] (out); accumulate != 0 adds into it.
 */
int td_smooth_bwd(const float* disp, const float* img, const float* mean,
                  int B, int h, int w, int normalize,
                  const float* gscale, float weight,
                  float* g_hat, float* dot_partial, float* d_disp, int accumulate,
                  td_stream_t stream);

/* Element types of the activation kernels below. */
#define TD_DTYPE_F32 0
#define TD_DTYPE_BF16 1

/*
 * nn.MaxPool2d(kernel_size=5, stride=1, padding=2) of the CRP blocks
 * (mono/model/mono_fm_joint/layers.py:208,213) on channels-last activations.
 *   in / out / grad_* : [N,H,W,C] (NHWC memory, i.e. a torch channels_last tensor), C % 8 == 0
 *   idx               : [N,H,W,C] uint8, window offset dy*5+dx of the selected element
 *                       (ATen's tie-break: first maximum in row-major order; NaN propagates)
 */
int td_maxpool5_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, uint8_t* idx,
                    td_stream_t stream);
int td_maxpool5_bwd(const void* grad_out, const uint8_t* idx, int dtype, int N, int H, int W, int C,
                    void* grad_in, td_stream_t stream);
/* The same with a tensor add [N,H,W,C] (nullable) summed into the result: grad_in = maxpool5_backward(grad_out) + add -- the gradient an
 * input of the CRP block's chain receives both through its pool and directly from the running sum (layers.py:200-215). */
int td_maxpool5_bwd_add(const void* grad_out, const uint8_t* idx, const void* add, int dtype, int N, int H, int W, int C,
                        void* grad_in, td_stream_t stream);

/*
 * The ResNet stem pool, nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (mono/model/mono_fm_joint/resnet.py:101),
 * same conventions as td_maxpool5_*: in [N,H,W,C] -> out / idx [N,Ho,Wo,C] with Ho = (H - 1) / 2 + 1,
 * idx = dy*3+dx of the selected element; H, W in td_maxpool3s2_bwd are the INPUT sizes.
 */
int td_maxpool3s2_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, uint8_t* idx,
                      td_stream_t stream);
int td_maxpool3s2_bwd(const void* grad_out, const uint8_t* idx, int dtype, int N, int H, int W, int C,
                      void* grad_in, td_stream_t stream);

/*
 * Channel concatenation of the DepthDecoder stages on channels-last activations
 * (mono/model/mono_fm_joint/depth_decoder.py:89-103, torch.cat((reduce(l), x, disp), 1)):
 *   out[pix, :] = [a[pix, :C0], b[pix, :C1], tail[pix, :C2], 0 ... 0]   with C0 + C1 + 8 output channels,
 *   C0 % 8 == C1 % 8 == 0, 1 <= C2 <= 8; all tensors [npix, C] row-major (NHWC), dtype f32 or bf16.
 * td_join_bwd writes the three slices of grad_out (the padding's gradient is dropped).
 */
int td_join_fwd(const void* a, const void* b, const void* tail, int dtype, long long npix, int C0, int C1, int C2,
                void* out, td_stream_t stream);
int td_join_bwd(const void* grad_out, int dtype, long long npix, int C0, int C1, int C2, void* grad_a, void* grad_b,
                void* grad_tail, td_stream_t stream);
/* The same with the middle operand `b_half` [N,H/2,W/2,C1] up-sampled x2 (nearest) on the fly: the DepthDecoder's
 * torch.cat((reduce(l), upsample(x), disp), 1) without materialising upsample(x) (depth_decoder.py:89-103; 189 MB at the
 * last stage of C2).  H, W even.  The backward sums the four output pixels of every low-resolution pixel. */
int td_join_up2_fwd(const void* a, const void* b_half, const void* tail, int dtype, int N, int H, int W, int C0, int C1,
                    int C2, void* out, td_stream_t stream);
int td_join_up2_bwd(const void* grad_out, int dtype, int N, int H, int W, int C0, int C1, int C2, void* grad_a,
                    void* grad_b_half, void* grad_tail, td_stream_t stream);

/*
 * Training-mode BatchNorm2d on channels-last activations with the residual add and ReLU of the ResNet
 * blocks fused in (mono/model/mono_fm_joint/resnet.py:30-49 BasicBlock.forward, :66-86
 * Bottleneck.forward; F.batch_norm(training=True) semantics: biased batch variance for the output,
 * unbiased for running_var, running = (1 - momentum) * running + momentum * batch).
 *
 *   x, residual, y, dy, dx, dresidual : [M, C] row-major (= NHWC with M = N*H*W), dtype f32 or bf16, C % 64 == 0
 *   gamma, beta, running_*, save_*, dgamma, dbeta : [C] f32 ; residual / running_* / dresidual may be NULL
 *   y = relu?( (x - mean) * invstd * gamma + beta [+ residual] )
 *   backward: g = dy * [y > 0] (relu; with y == NULL and no residual the mask is recomputed from x, gamma, beta,
 *             save_mean, save_invstd exactly as the forward formed it) ; dbeta = sum g ; dgamma = sum g * xhat ;
 *             dx = gamma * invstd * (g - dbeta / M - xhat * dgamma / M) ; dresidual = g
 *   groups: the M rows are `groups` consecutive equal ranges with separate batch statistics (save_mean /
 *           save_invstd are [groups, C]); the running statistics receive one momentum update per group, in
 *           order, and dgamma / dbeta are summed over the groups -- i.e. exactly `groups` separate calls on the
 *           stacked passes of one network over different frames (pose pairs, source-frame features).
 *   workspace: td_bn_workspace_floats(M, groups, C) floats (partial sums + coefficients; contents undefined on return).
 */
long long td_bn_workspace_floats(long long M, int groups, int C);
int td_bn_fwd(const void* x, const void* residual, int dtype, const float* gamma, const float* beta,
              float* running_mean, float* running_var, float momentum, float eps, int relu, long long M, int groups,
              int C, void* y, float* save_mean, float* save_invstd, float* workspace, td_stream_t stream);
int td_bn_bwd(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta, const float* save_mean,
              const float* save_invstd, int relu, long long M, int groups, int C, void* dx, void* dresidual,
              float* dgamma, float* dbeta, float* workspace, td_stream_t stream);

/*
 * 1x1 convolution of the ResNet bottlenecks on channels-last bf16 activations as an MFMA GEMM
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulation) whose epilogue forms the partial batch statistics of the
 * BatchNorm that follows it.  Replaces nn.Conv2d(kernel_size=1) + the statistics pass of nn.BatchNorm2d in
 * Bottleneck.forward (mono/model/mono_fm_joint/resnet.py:66-86: conv1 -> bn1, conv3 -> bn3) and in the strided
 * 1x1 down-sample branch (resnet.py:119-127).
 *   x [rows_in, K] bf16 (rows_in = M for stride 1; N_img*Hi*Wi for the strided form), w [N, K] bf16 (the [N,K,1,1] weight),
 *   y [M, N] bf16, M = output pixels; K % 64 == 0, N % 64 == 0; groups: M is `groups` consecutive equal row ranges with
 *   separate statistics (stacked passes, as td_bn_fwd).
 *   stat_partials (may be NULL): [groups, S, N, 2] f32 with S = td_conv1x1_stat_rows(M, groups, N): (sum y, sum y^2) over the
 *   rows of each row tile, of the bf16-rounded outputs -- the layout td_bn_fwd_from_partials consumes.
 * td_bn_fwd_from_partials: td_bn_fwd without its statistics pass: mean / invstd / running statistics from `partials`
 *   [groups, stat_rows, C, 2] (finished in the prologue of the apply kernel; the buffer is SCRATCH: more than 96 rows are first
 *   reduced in place), then y = relu?( (x - mean) * invstd * gamma + beta [+ residual] ).
 * td_conv1x1_wgrad: the weight gradient of the same convolution, dW[n, k] = sum_m dY[m, n] * X[src(m), k] (autograd of the
 *   Conv2d above), on the MFMA through transposed LDS reads (ds_read_b64_tr_b16); M is cut into row ranges whose fp32 partial
 *   tiles a second kernel adds in order (deterministic; dW in bf16 or f32: dw_dtype).
 *   workspace: td_conv1x1_wgrad_workspace_floats(M, K, N) floats.
 */
int td_conv1x1_stat_rows(long long M, int groups, int N);
int td_conv1x1_fwd(const void* x, const void* w, long long M, int groups, int K, int N, int Hi, int Wi, int stride, void* y,
                   float* stat_partials, td_stream_t stream);
/* 3x3 convolution (stride 1 or 2, padding 0 or 1) of the ResNet blocks and the decoders as an implicit MFMA GEMM: K = 9 * Cin
 * walked tap by tap over the channels-last input (the im2col matrix is never formed), same epilogue as td_conv1x1_fwd
 * (stat_partials: td_conv1x1_stat_rows(B*Ho*Wo, groups, N) row tiles; may be NULL).  Replaces conv2 (+ bn2's statistics pass)
 * of Bottleneck.forward and the convolutions of BasicBlock.forward (mono/model/mono_fm_joint/resnet.py:30-49, 66-86).
 *   x [B, Hi, Wi, Cin] bf16 channels-last, w [N, 3, 3, Cin] bf16 (the memory of a channels-last [N, Cin, 3, 3] weight),
 *   y [B, Ho, Wo, N] bf16; Cin % 8 == 0, N % 64 == 0. */
int td_conv3x3_fwd(const void* x, const void* w, int B, int groups, int Hi, int Wi, int Cin, int N, int stride, int pad, void* y,
                   float* stat_partials, td_stream_t stream);
/* Weight gradient of a 3x3 stride-1 convolution (zero padding `pad` = 1: conv2 of the ResNet blocks, mono/model/mono_fm_joint/resnet.py:
 * 30-49, 57-58; `pad` = 0 on a pre-padded input: Conv3x3 = ReflectionPad2d(1) + conv, mono/model/mono_fm_joint/layers.py:171-184):
 *   dW[n, ky, kx, c] = sum_(b,ho,wo) dY[b, ho, wo, n] * X[b, ho + ky - pad, wo + kx - pad, c]
 *   dy [B, Ho, Wo, N], x [B, Ho + 2 - 2 pad, Wo + 2 - 2 pad, C] bf16 channels-last; dw [N, 3, 3, C] (the memory of a channels-last
 *   [N, C, 3, 3] weight) in bf16 or f32 (dw_dtype); C % 64 == N % 64 == 0, Wo >= 20, Ho >= 4.
 * MFMA over the pixel index like td_conv1x1_wgrad, nine taps per staged tile, fp32 slabs per row range added in a fixed order
 * (deterministic; MIOpen's split-K kernels accumulate with atomics into a zero-filled fp32 workspace and cast afterwards).
 *   workspace: td_conv3x3_wgrad_workspace_floats(B, Ho, Wo, C, N) floats. */
long long td_conv3x3_wgrad_workspace_floats(int B, int Ho, int Wo, int C, int N);
int td_conv3x3_wgrad(const void* dy, const void* x, int B, int Ho, int Wo, int C, int N, int pad, int dw_dtype, void* dw,
                     float* workspace, td_stream_t stream);
long long td_conv1x1_wgrad_workspace_floats(long long M, int K, int N);
int td_conv1x1_wgrad(const void* dy, const void* x, long long M, int K, int N, int Hi, int Wi, int stride, int dw_dtype, void* dw,
                     float* workspace, td_stream_t stream);
/* n weight gradients in few launches: problem i == td_conv1x1_wgrad(dy[i], x[i], M[i], K[i], N[i], Hi[i], Wi[i], stride[i], dw_dtype[i],
 * dw[i], workspace[i]), bit for bit (same tiling / row ranges / summation order); all array arguments are HOST arrays of length n
 * (<= 4096).  A training step has ~90 independent 1x1 weight gradients that nothing needs before the optimiser step: enqueued by
 * the autograd nodes and launched together they stop being ~180 latency-bound launches (tripled_amd.ops.deferred_wgrads). */
int td_conv1x1_wgrad_group(int n, const void* const* dy, const void* const* x, const long long* M, const int* K, const int* N,
                           const int* Hi, const int* Wi, const int* stride, const int* dw_dtype, void* const* dw,
                           float* const* workspace, td_stream_t stream);
int td_bn_fwd_from_partials(const void* x, const void* residual, int dtype, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, int relu, long long M,
                            int groups, int C, float* partials, int stat_rows, void* y, float* save_mean,
                            float* save_invstd, td_stream_t stream);

/*
 * Bias + activation behind a convolution and its adjoint with the bias gradient, on channels-last activations [M, C] (f32 or bf16,
 * C % 8 == 0, C <= 256, 256 % (C / 8) == 0).  Replaces, around the decoders' convolutions (ConvBlock = Conv3x3 -> ELU,
 * mono/model/mono_fm_joint/layers.py:143-155; F.leaky_relu(iconv / merge), mono/model/mono_fm_joint/depth_decoder.py:89-103): the bias
 * add of the convolution, the activation, the activation's backward and the bias-gradient reduction of the convolution's backward.
 *   act: 0 none, 1 ELU(alpha = 1), 2 leaky ReLU(0.01);  bias [C] f32 or bf16 (bias_dtype; NULL = no bias)
 *   td_bias_act_fwd: a = act(y + bias)            (a may alias y)
 *   td_bias_act_bwd: gy = g * act'(a) (from the RESULT a; gy may alias g; act 0: gy is not written, dbias = column sums of g),
 *                    dbias [C] (nullable) = sum over rows of gy, deterministic; workspace: td_bias_act_workspace_floats(M, C) floats.
 */
long long td_bias_act_workspace_floats(long long M, int C);
int td_bias_act_fwd(const void* y, const void* bias, int bias_dtype, int dtype, long long M, int C, int act, void* a, td_stream_t stream);
int td_bias_act_bwd(const void* g, const void* a, int dtype, long long M, int C, int act, void* gy, void* dbias, int dbias_dtype,
                    float* workspace, td_stream_t stream);

/*
 * Optimiser step of a flat fp32 parameter buffer in one pass: the gradient scale of torch.nn.utils.clip_grad_norm_ (total_norm: device
 * scalar, the 2-norm of grad, or NULL for no clipping), torch.optim.Adam's update (weight_decay 0, no amsgrad; `step` = device scalar
 * holding the 1-based step count, lr from the device scalar lr_dev or, when NULL, lr_host) and the bf16 working copy of the first
 * n_lowp parameters.  Replaces optimizer_config.grad_clip + optimizer.step() of the training hook (mono/core/utils/dist_utils.py:54-60)
 * on the flat store of tripled_amd/flat_amp.py.  n % 4 == 0, n_lowp % 4 == 0.
 */
int td_adam_flat(float* w, const float* grad, float* exp_avg, float* exp_avg_sq, void* lowp, long long n, long long n_lowp,
                 const float* step, const float* lr_dev, float lr_host, double beta1, double beta2, float eps, const float* total_norm,
                 float max_norm, td_stream_t stream);

/*
 * flat[dst_offsets[i] + e] = (float) srcs[i][e] for e < numels[i], i < n; srcs[i] == NULL zero-fills the slot.  srcs / dst_offsets /
 * numels are HOST arrays (the pointers travel in the kernel arguments, 64 tensors per launch: legal under graph capture); all
 * sources have the dtype src_dtype (bf16 or f32).  The gradient gather of the flat parameter store (tripled_amd/flat_amp.py):
 * replaces torch.cat over the per-parameter gradients + the bf16 -> f32 pass.
 */
int td_gather_flat(const void* const* srcs, const long long* dst_offsets, const long long* numels, int n, int src_dtype, float* flat,
                   td_stream_t stream);

/*
 * Fused forms of the bottleneck's 1x1 convolutions (round 4): the BatchNorm passes of the NEIGHBOURING layers ride on the GEMM's
 * operand staging and epilogue instead of being separate passes over the activations.  Reference: Bottleneck.forward,
 * mono/model/mono_fm_joint/resnet.py:66-86 (conv1 -> bn1 -> relu -> conv2 -> bn2 -> relu -> conv3 -> bn3 -> += identity -> relu)
 * and what autograd derives from it.  All activations [M, C] bf16 row-major (channels-last), stride 1, channel counts % 64 == 0,
 * `groups` stacked passes with separate statistics as td_bn_fwd.
 *
 * td_bn_partial_rows / td_bn_fwd_partials / td_bn_bwd_partials: the statistics passes of td_bn_fwd / td_bn_bwd on their own:
 *   partials [groups, td_bn_partial_rows(M, groups, C), C, 2] f32 = per row range (sum x, sum x^2), resp. (sum g, sum g (x - mean))
 *   with g = dy masked by the ReLU exactly as td_bn_bwd does.  The buffer is SCRATCH for the consumer (reduced in place).
 * td_bn_bwd_from_partials: td_bn_bwd without its statistics pass (the sums come from td_conv1x1_dgrad_bnsums' epilogue).
 *
 * td_conv1x1_fwd_bnrelu: y = conv1x1(relu(bn(z)), w) where bn's batch statistics come from `in_partials` (in_rows rows per group, e.g.
 *   from td_bn_fwd_partials; finished in the GEMM's prologue, which also writes save_mean / save_invstd [groups, K] and updates the
 *   running statistics): conv3(relu(bn2(conv2 output))) of the bottleneck without the bn2 pass.  a_side (nullable) [M, K] receives
 *   relu(bn(z)) (the weight gradient of this convolution needs it); stat_partials as td_conv1x1_fwd.  K <= 512.
 * td_conv1x1_dgrad: dx[M, Cin] = dy[M, Cout] . w[Cout, Cin] (+ residual [M, Cin], added to the bf16-rounded product and rounded
 *   again: the tensor add autograd runs behind the data gradient for the block's identity branch).  The weight is read as the
 *   forward stores it (transposed LDS reads).
 * td_conv1x1_dgrad_bnsums: the same GEMM whose epilogue forms the backward sums of the BatchNorm that PRODUCED this convolution's
 *   input: out_partials [groups, td_conv1x1_stat_rows(M, groups, Cin), Cin, 2] = (sum g, sum g (z - mean)), g = dx * [bn(z) > 0],
 *   z [M, Cin] that BatchNorm's input (conv3's data gradient + bn2's backward statistics, no pass over dx for them).
 * td_conv1x1_dgrad_bnbwd: dx = BNbackward(g, z) . w (+ residual): the dy operand is formed while it is staged from the gradient g
 *   [M, Cout] of relu(bn(z)) and z [M, Cout] (dz = gamma invstd (g' - mean(g') - xhat mean(g' xhat)), g' = g [bn(z) > 0]); the sums
 *   come from in_partials (td_bn_bwd_partials), dgamma / dbeta [Cout] are written, dz_side (nullable) [M, Cout] receives dz for the
 *   weight gradient: conv1's data gradient straight from conv2's, without the bn1 dx pass and without the residual add.  Cout <= 512.
 */
int td_bn_partial_rows(long long M, int groups, int C);
int td_bn_fwd_partials(const void* x, int dtype, long long M, int groups, int C, float* partials, td_stream_t stream);
int td_bn_bwd_partials(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C, float* partials,
                       td_stream_t stream);
int td_bn_bwd_from_partials(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                            const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                            float* partials, int stat_rows, void* dx, void* dresidual, float* dgamma, float* dbeta,
                            td_stream_t stream);
int td_conv1x1_fwd_bnrelu(const void* z, const void* w, long long M, int groups, int K, int N, float* in_partials, int in_rows,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                          float* save_mean, float* save_invstd, void* a_side, void* y, float* stat_partials, td_stream_t stream);
int td_conv1x1_dgrad(const void* dy, const void* w, long long M, int groups, int Cout, int Cin, const void* residual, void* dx,
                     td_stream_t stream);
/* y = conv1x1(x, w) [M, N] and run_out = run_in + y (both bf16 [M, N]; run_out rounded from the bf16 y + run_in, as the tensor add
 * would be): the pointwise convolution of a CRP stage with the block's running sum in its epilogue (layers.py:200-215). */
int td_conv1x1_fwd_sum(const void* x, const void* w, long long M, int K, int N, const void* run_in, void* y, void* run_out,
                       td_stream_t stream);
int td_conv1x1_dgrad_bnsums(const void* dy, const void* w, long long M, int groups, int Cout, int Cin, const void* z,
                            const float* gamma, const float* beta, const float* save_mean, const float* save_invstd, void* dx,
                            float* out_partials, td_stream_t stream);
int td_conv1x1_dgrad_bnbwd(const void* g, const void* z, const void* w, long long M, int groups, int Cout, int Cin, float* in_partials,
                           int in_rows, const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                           float* dgamma, float* dbeta, void* dz_side, const void* residual, void* dx, td_stream_t stream);

/*
 * The same normalisation with statistics synchronised over the data-parallel ranks (the reference trains with
 * syncbn=True: torch.nn.SyncBatchNorm.convert_sync_batchnorm, mono/apis/trainer.py:156-157).  The caller
 * all-reduces (SUM) the [groups, C, 2] per-channel sums and the row count between the two stages; everything
 * heavy stays in the kernels of td_bn_fwd / td_bn_bwd:
 *   td_bn_sync_fwd_sums   sums[g,c,:] = (sum x, sum x^2) over this rank's rows
 *   td_bn_sync_fwd_apply  mean / invstd / running statistics from the GLOBAL sums and `count` (device scalar:
 *                         global rows per group), then y as in td_bn_fwd
 *   td_bn_sync_bwd_sums   sums[g,c,:] = (sum g, sum g * (x - mean)),  g = dy masked by the ReLU
 *   td_bn_sync_bwd_dx     dgamma / dbeta from the LOCAL sums (the gradient all-reduce averages them like every
 *                         parameter gradient), dx from the GLOBAL sums; coef: [groups, C, 3] scratch
 *   workspace: td_bn_workspace_floats(M, groups, C) floats.
 */
int td_bn_sync_fwd_sums(const void* x, int dtype, long long M, int groups, int C, float* sums, float* workspace,
                        td_stream_t stream);
int td_bn_sync_fwd_apply(const void* x, const void* residual, int dtype, const float* sums, const float* count,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                         float eps, int relu, long long M, int groups, int C, void* y, float* save_mean,
                         float* save_invstd, td_stream_t stream);
int td_bn_sync_bwd_sums(const void* dy, const void* x, const void* y, int dtype, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                        float* sums, float* workspace, td_stream_t stream);
int td_bn_sync_bwd_dx(const void* dy, const void* x, const void* y, int dtype, const float* local_sums,
                      const float* global_sums, const float* count, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, int relu, long long M, int groups, int C,
                      void* dx, void* dresidual, float* dgamma, float* dbeta, float* coef, td_stream_t stream);

/*
 * Edge-aware regulariser on C-channel feature maps: get_feature_regularization_loss,
 * mono/model/mono_fm_joint/net.py:309-330 (six stencil terms |d_k F| * exp(-a * mean_c |d_k I|)).
 *
 * td_edge_weights: Wt[b,k,y,x] = scale6[k] * exp(-a * mean_c |d_k img|), 0 where term k has no anchor
 *   at (y,x); k = dx, dy, dxx, dxy, dyx, dyy.  img [B,3,h,w] is the area-resized target; scale6 (HOST
 *   array) carries coefficient / element-count of each term, so that
 *       loss = sum over anchors, channels, k of |d_k F| * Wt
 * td_featreg_fwd: partial[td_featreg_num_blocks] per-block sums of that (finish with td_sum_scaled).
 * td_featreg_bwd: grad = gscale[0] * d loss / d feat, same dtype/layout as feat.
 *   feat: [B,h,w,C] channels-last memory, C % 8 == 0, dtype TD_DTYPE_F32 or TD_DTYPE_BF16.
 */
int td_edge_weights(const float* img, int B, int h, int w, float a, const float* scale6, float* Wt,
                    td_stream_t stream);
int td_featreg_num_blocks(int B, int h, int w, int C);
int td_featreg_fwd(const void* feat, int dtype, const float* Wt, int B, int h, int w, int C,
                   float* partial, td_stream_t stream);
int td_featreg_bwd(const void* feat, int dtype, const float* Wt, const float* gscale, int B, int h, int w,
                   int C, void* grad, td_stream_t stream);

/*
 * Masked photometric reconstruction term of the in-painting auto-encoder
 * (mono/model/mono_fm_joint_inpaint/net.py:80-91):
 *     S = sum_p hole[p] * (0.85 * mean_c SSIM_c(x, y)(p) + 0.15 * mean_c sqrt((y - x)^2 + 1e-6))
 *   x (prediction), y (target): [B,3,h,w] f32 planar; hole: [B,h,w] f32 per-pixel weight
 * td_recon_fwd: partial[td_recon_num_tasks] per-wave sums of S (finish with td_sum_scaled);
 * td_recon_bwd: dx = gscale[0] * dS/dx, [B,3,h,w].
 */
int td_recon_num_tasks(int B, int h, int w);
int td_recon_fwd(const float* x, const float* y, const float* hole, int B, int h, int w,
                 float* partial, td_stream_t stream);
int td_recon_bwd(const float* x, const float* y, const float* hole, const float* gscale, int B, int h,
                 int w, float* dx, td_stream_t stream);

/*
 * nn.ReflectionPad2d(1) in front of every decoder Conv3x3 (mono/model/mono_fm_joint/layers.py:171-184) on
 * channels-last activations: in [N,H,W,C] -> out [N,H+2,W+2,C]; the backward is the gather-form adjoint
 * (grad_out [N,H+2,W+2,C] -> grad_in [N,H,W,C]).  C % 8 == 0, H, W >= 2.
 */
int td_reflpad1_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, td_stream_t stream);
int td_reflpad1_bwd(const void* grad_out, int dtype, int N, int H, int W, int C, void* grad_in,
                    td_stream_t stream);

/*
 * out = ReflectionPad2d(1)(F.interpolate(in, scale_factor=2, mode="nearest")) without materialising the
 * up-sampled tensor: the `iconv(upsample(upconv(x)))` step of the image decoders
 * (mono/model/mono_fm_joint/decoder.py:40-57).  in [N,H,W,C], out [N,2H+2,2W+2,C], channels-last, C % 8 == 0.
 * td_up2_reflpad1_bwd is the exact adjoint (gather form, <= 16 taps at a corner, 4 in the interior).
 */
int td_up2_reflpad1_fwd(const void* in, int dtype, int N, int H, int W, int C, void* out, td_stream_t stream);
int td_up2_reflpad1_bwd(const void* grad_out, int dtype, int N, int H, int W, int C, void* grad_in,
                        td_stream_t stream);

/*
 * Feature-metric term: generate_features_pred (mono/model/mono_fm_joint/net.py:196-223) +
 * compute_perceptional_loss (:63-65) + min over source frames (mono_fm_joint_inpaint/net.py:58-70), fused.
 *   tgt, src[i]  [B,h,w,C] channels-last feature maps (C % 64 == 0; dtype f32 or bf16); n_src <= 2
 *   disp         [B,1,hs,ws]; P [n_src,B,3,4] and invK [B,4,4] at FEATURE resolution (K rows 0,1 halved)
 *   argmin       [B,h,w] uint8 (out: which source frame gives the minimum)
 *   partial      [td_featwarp_num_blocks] per-block sums of min_f mean_c sqrt((tgt - warp_f)^2 + 1e-6)
 * Backward (gradient of gscale[0] * inv_count * sum(partial)):
 *   d_tgt [B,h,w,C] (feature dtype), d_src[i] [B,h,w,C] int32 fixed-point accumulators -- MUST be zero-filled by the caller; the
 *   scatter adds contributions rounded to multiples of |gscale| * inv_count / C / 2^14 with integer atomics (order-independent,
 *   bit-reproducible); td_featwarp_dsrc_finish turns them into the gradient in the feature dtype --,
 *   d_up [B,h,w] (w.r.t. the up-sampled disparity; td_upsample_adjoint),
 *   dP_partial [td_featwarp_num_blocks, n_src*12] (td_reduce_partials with blocks_per_sample =
 *   td_featwarp_num_blocks / B).
 */
int td_featwarp_num_blocks(int B, int h, int w);
int td_featwarp_fwd(const void* tgt, const void* const* src, int n_src, int dtype, const float* disp,
                    const float* P, const float* invK, int B, int h, int w, int C, int hs, int ws,
                    float min_depth, float max_depth, uint8_t* argmin, float* partial, td_stream_t stream);
int td_featwarp_bwd(const void* tgt, const void* const* src, int n_src, int dtype, const float* disp,
                    const float* P, const float* invK, const uint8_t* argmin, const float* gscale,
                    float inv_count, int B, int h, int w, int C, int hs, int ws, float min_depth,
                    float max_depth, void* d_tgt, int* const* d_src, float* d_up, float* dP_partial,
                    td_stream_t stream);
/* out[i] = acc[i] * |gscale[0]| * inv_count / C / 2^14 (n elements, n % 8 == 0; dtype f32 or bf16): the scatter's accumulators
 * -> d/d(source features). */
int td_featwarp_dsrc_finish(const int* acc, const float* gscale, float inv_count, int C, long long n, int dtype, void* out,
                            td_stream_t stream);
/* dP[i,b,:] = sum over the blocks_per_sample consecutive rows of partial[., n_src*12] that belong to sample b. */
int td_reduce_partials(const float* partial, int n_src, int B, int blocks_per_sample, float* dP,
                       td_stream_t stream);

/*
 * Robust-L1 channel-mean map between a network output and an image, and its adjoint:
 *   out[b,y,x] = weight * mean_c sqrt((pred[b,c,y,x] - target[b,c,y,x])^2 + 1e-6)
 * Replaces compute_perceptional_loss on images -- compute_auto_res_loss,
 * mono/model/mono_fm_joint_inpaint/net.py:520-527 (a per-pixel MAP that batch_processor's mean reduces), and
 * compute_colorization_loss, :310-323 -- with robust_l1 of mono/model/mono_fm_joint/net.py:59-65.
 *   pred          C-channel tensor of dtype TD_DTYPE_F32 / TD_DTYPE_BF16 with arbitrary element strides
 *   pred_strides  HOST array of 4 element strides (n, c, h, w): channels-last decoder outputs and channel
 *                 slices are read in place, no .float()/.contiguous() copy
 *   target        [B,C,H,W] fp32 NCHW;  out [B,1,H,W] fp32
 * td_l1map_bwd: dpred (same dtype and strides as pred) = gmap[b,y,x] * weight / C * (pred - target) / sqrt(...).
 */
int td_l1map_fwd(const void* pred, int dtype, const long long* pred_strides, const float* target, int B, int C,
                 int H, int W, float weight, float* out, td_stream_t stream);
int td_l1map_bwd(const void* pred, int dtype, const long long* pred_strides, const float* target,
                 const float* gmap, int B, int C, int H, int W, float weight, void* dpred, td_stream_t stream);

/*
 * sRGB -> normalised CIE Lab: rgb2lab, mono/model/mono_fm_joint_inpaint/color_conversions.py:106-114 (rgb2xyz
 * :6-27, xyz2lab :52-75), called on the target frame by the colourisation models (net.py:236,292).
 *   rgb [B,3,H,W] fp32 in [0,1];  lab [B,3,H,W] = ((L - l_cent) / l_norm, a / ab_norm, b / ab_norm).
 * No gradient (the input is data).
 */
int td_rgb2lab(const float* rgb, int B, int H, int W, float l_cent, float l_norm, float ab_norm, float* lab,
               td_stream_t stream);

/*
 * Pose vectors -> camera transforms -> projection matrices for all frame pairs of a step in one launch.
 * Replaces transformation_from_parameters / rot_from_axisangle / get_translation_matrix,
 * mono/model/mono_fm_joint/net.py:225-277, and torch.matmul(K, T)[:, :3, :] of Project.forward,
 * mono/model/mono_fm_joint/layers.py:73-75.
 *   axisangle, translation  [n_pairs*B, 3] (PoseDecoder outputs, pair-major)
 *   invert                  HOST array of n_pairs flags (frame id < 0: M = R^T Trans(-t), else Trans(t) R)
 *   K                       [B,4,4] intrinsics (may be NULL when P is NULL)
 *   T   [n_pairs,B,4,4] (out)     P  [n_pairs,B,3,4] = (K @ T)[:, :3, :] (out, may be NULL)
 * td_pose_bwd: g_axisangle / g_translation [n_pairs*B,3] from gT and/or gP (either may be NULL, not both).
 */
int td_pose_fwd(const float* axisangle, const float* translation, const int* invert, const float* K,
                int n_pairs, int B, float* T, float* P, td_stream_t stream);
int td_pose_bwd(const float* axisangle, const float* translation, const int* invert, const float* K,
                int n_pairs, int B, const float* gT, const float* gP, float* g_axisangle,
                float* g_translation, td_stream_t stream);

/*
 * Input expansion on the device: uint8 frames -> float images + their colour-jittered copies.  Replaces ToTensor and
 * ColorJitter per frame on the host (MonoDataset.preprocess, mono/datasets/mono_dataset.py:83-101; parameters drawn at
 * :146-152) and the float32 upload of both copies (change_input_variable, mono/apis/trainer.py:19-29).
 *   frames_u8  [N,3,H,W] uint8 (N = frames x samples)
 *   aug        [N,9] float: (enabled, op0..op3, brightness, contrast, saturation, hue); ops 0..3 = brightness, contrast,
 *              saturation, hue, applied in the listed order (torchvision ColorJitter's float formulas)
 *   means_scratch [N] floats;  color, color_aug [N,3,H,W] float32 in [0,1] (out)
 */
int td_color_jitter(const uint8_t* frames_u8, const float* aug, int N, int H, int W, float* means_scratch, float* color,
                    float* color_aug, td_stream_t stream);

/*
 * Per-tensor fp8 quantisation (OCP e4m3fn, saturating) for the fp8 1x1-convolution path (BASELINE config 5).  The
 * reference has no reduced-precision path (it trains in fp32: mono/apis/trainer.py:147-189); the 1x1 convolutions this
 * serves are resnet.py:52-86 (Bottleneck conv1/conv3, downsample) and layers.py:110-118 (Conv1x1).
 *   td_fp8_amax_partials  partials[td_fp8_num_blocks(n)] = per-block max |x|
 *   td_fp8_quantize       q[i] = e4m3(x[i] * 448 / amax), inv_scale[0] = amax / 448  (amax = max of the partials; the
 *                         product  q_a . q_b * inv_scale_a * inv_scale_b  is the de-quantised GEMM result)
 *   x: n elements of dtype TD_DTYPE_F32 / TD_DTYPE_BF16, n % 8 == 0.
 */
int td_fp8_num_blocks(long long n);
int td_fp8_amax_partials(const void* x, int dtype, long long n, float* partials, td_stream_t stream);
int td_fp8_quantize(const void* x, int dtype, long long n, const float* partials, uint8_t* q, float* inv_scale,
                    td_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIPLED_HIP_H */
