/*
 * pam.h -- C ABI of libpam_hip.so: the MI355X (gfx950) implementation of the per-frame multi-view 3D pose
 * hot path of B10532021/Part-Aware_Measurement_for_3D_Pose_Estimation_and_Tracking.
 *
 * The reference has no FFI (pure Python); the seams this ABI replaces are the Python call sites below
 * (paths relative to /root/reference/src).  Host code (the package's ivclabpose.py / tracker.py) binds these with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns 0 on success or a negative PAM_E_*
 * code, with text available from pam_last_error(); no exceptions or callbacks cross the ABI; a handle owns its
 * device memory, is bound to one GPU and is not thread-safe.  "host" pointers are ordinary host memory, "dev"
 * pointers are device memory (e.g. torch tensor .data_ptr()); `stream` is a hipStream_t passed as void* (NULL =
 * the legacy null stream, which is what torch's default stream is).  Keypoints are rows (y, x, score) in float64 -- the tracker-internal layout the
 * reference builds at ivclabpose.py:236-244.
 */
#ifndef PAM_H
#define PAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAM_J 17            /* joints; hard-coded 17 in the reference too (ivclabpose.py:96) */
#define PAM_MAX_VIEWS 32    /* per-joint view sets are 32-bit masks */
#define PAM_MAX_TAPS 16
#define PAM_EXP_TABLE 64
#define PAM_YOLO_MAX_CAND 1024 /* boxes above the score threshold kept per image before NMS */

#define PAM_OK 0
#define PAM_E_ARG (-1)      /* bad argument / capacity */
#define PAM_E_HIP (-2)      /* HIP runtime error */
#define PAM_E_STATE (-3)    /* call order (e.g. cameras not set) */
#define PAM_E_OVERFLOW (-4) /* a scene ran out of track / hypothesis slots (see out status word) */

/* Matcher hyper-parameters: ivclabpose.py:140-156 (iter_args) + derived tables computed by the host exactly as
 * the reference's Python does (math.exp / scipy _gaussian_kernel1d), so the device uses bit-identical constants. */
typedef struct PamParams {
    double conf_threshold;       /* PIPELINE_COMBINATION.CONF_THRESHOLD            */
    double epi_threshold;        /* EPI_THRESHOLD   (hypothesis cost scale)         */
    double init_threshold;       /* INIT_THRESHOLD  (init-path joint filter, f32)   */
    double joint_threshold;      /* JOINT_THRESHOLD (update-path joint filter)      */
    double alpha2d;              /* ALPHA2D                                         */
    double lambda_a;             /* LAMBDA_A                                        */
    double lambda_t;             /* LAMBDA_T                                        */
    int32_t n_init;              /* N_INIT                                          */
    int32_t max_age;             /* MAX_AGE                                         */
    int32_t count_gate;          /* 10: the literal at IterativeTracker.py:145      */
    int32_t n_taps_body;         /* radius+1 taps of gaussian_filter1d(sigma)       */
    int32_t n_taps_arm;          /* radius+1 taps of gaussian_filter1d(arm_sigma)   */
    int32_t reserved;
    double taps_body[PAM_MAX_TAPS];   /* w[0]=centre ... w[r]                       */
    double taps_arm[PAM_MAX_TAPS];
    double exp_lambda_a[PAM_EXP_TABLE]; /* exp(lambda_a * dt), dt = 0..63           */
    double w_lambda_t[4];               /* exp(-lambda_t * T), T = 0..3             */
} PamParams;

typedef struct PamHandle PamHandle;

/* Layout of the per-scene output record (int32 words then float64 words); filled by pam_out_layout(). */
typedef struct PamOutLayout {
    int32_t n_views, max_dets, max_tracks, n_scenes;
    int32_t int_words;      /* int32 words per scene  */
    int32_t dbl_words;      /* float64 words per scene */
    /* int32 section: header then per-track blocks (list order = the reference's self.tracks order) */
    int32_t hdr_words;      /* [0]=n_tracks [1]=status word: bits 0-15 = status bits OF THIS FRAME: 1 = out of track slots, 2 = out of
                             * hypothesis slots, 4 = an assignment problem was infeasible, 8 = a device-side detection count outside
                             * [0, max_dets] was clamped (pam_frame_dev), 16 = the frame's input was VOID (pam_set_input_guard): the frame was
                             * not applied -- state untouched, no tracks in the record, [3] = first frame of this run of void frames (never
                             * sticky); bits 16-31 = bits 1-8 OR-ed over EVERY frame since
                             * pam_create / pam_reset (sticky: a host that decodes only the last record of a run still sees them)
                             * [2]=frame_id [3]=n_hyp (debug) */
    int32_t trk_words;      /* words per track block */
    /* track block: 0 id,1 state,2 hits,3 age,4 time_since_update,5 emitted,6 n_2d_views,7 V (views offered to the
     * newest pose),8 history length,9 newest pose time, then order[n_views], matched_det[n_views] (index of the
     * detection matched THIS frame per camera id or -1), time2d[n_views] (per camera id), nviews[17] */
    int32_t off_order, off_matched, off_time2d, off_nviews;
    /* float64 section: [0..15] in-kernel clocks (s): [0] start, [1] after association, [2] after update, [3] after init,
     * [4..11] finer phase stamps (view selection, conflicts, DLT, success, smoothing, append, end of init, end of record);
     * then per track pose3d[17*3], velocity[17*3] */
    int32_t dbl_hdr_words;
    int32_t dbl_trk_words;
} PamOutLayout;

/* ---- life cycle --------------------------------------------------------------------------------------------
 * replaces: ivclabpose.__init__ (ivclabpose.py:136-158) + IterativeTracker.__init__ (IterativeTracker.py:36-45).
 * n_scenes > 1 runs that many independent tracker instances per launch (batched scenes, same rig). */
int pam_create(PamHandle** out, int device, int n_views, int max_dets, int max_tracks, int max_hyps,
               int n_scenes, const PamParams* params);
int pam_destroy(PamHandle* h);
const char* pam_last_error(const PamHandle* h);   /* h may be NULL: last create error */
const char* pam_version(void);

/* replaces: ivclabpose.GetCameraParameters / Camera.__init__ results (ivclabpose.py:35-46,162-181).  The host computes
 * P, F, RK_INV (float32) and position (float64) exactly as the reference does and hands them over. */
int pam_set_cameras(PamHandle* h, const float* P /*C*3*4*/, const float* F /*C*C*3*3*/,
                    const float* RK_INV /*C*3*3*/, const double* position /*C*3*/);

/* replaces: IterativeTracker.track_restart (IterativeTracker.py:47-50) */
int pam_reset(PamHandle* h);

int pam_out_layout(const PamHandle* h, PamOutLayout* out);

/* ---- the per-frame step ------------------------------------------------------------------------------------
 * replaces: IterativeTracker.tracking (IterativeTracker.py:115-180) + output collection (ivclabpose.py:259-287):
 * association, per-track part-aware view filter + weighted DLT + smoothing + motion, greedy hypothesis
 * initialisation, life cycle.  State stays on the device.
 *   n_det : n_scenes*C int32         detections per view
 *   det   : n_scenes*C*max_dets*17*3 float64 rows (y, x, score)
 * pam_frame: host buffers in, host record out (out_i: n_scenes*int_words, out_d: n_scenes*dbl_words); synchronous.
 * pam_frame_dev: device buffers in, record left in the handle's device output buffer; asynchronous on `stream`.
 * pam_fetch: async copy of the device output record to (pinned) host memory on `stream`. */
int pam_frame(PamHandle* h, int frame_id, const int32_t* n_det, const double* det, int32_t* out_i, double* out_d);
int pam_frame_dev(PamHandle* h, void* stream, int frame_id, const int32_t* dev_n_det, const double* dev_det);
/* the same step on VIEW-SHARDED input, read in place (no unpack kernel, no copies): dev_records = the buffer the all-gather of the
 * ranks' send buffers filled (pam_allgather_keypoints' dev_recv; record layout there), dev_view_row[v] (n_views int32) = which of its
 * records holds view v.  One scene per handle. */
int pam_frame_dev_views(PamHandle* h, void* stream, int frame_id, const double* dev_records, const int32_t* dev_view_row);
int pam_fetch(PamHandle* h, void* stream, int32_t* host_out_i, double* host_out_d);
int pam_sync(PamHandle* h, void* stream);
/* Input guard of the frame step (round 6).  The reference's PersonPoseDetect returns host lists before PersonTrack_Project3DPose runs
 * (/root/reference/src/ivclabpose.py:208-287): a frame can never be tracked on keypoints their producer disowns.  Here the tracker consumes
 * the decode on the device, so the producer's verdict has to be readable there: while *dev_word != 0 (a device int32 the producer raises:
 * pam_flag_gate's dev_void; for view-sharded input also the second double of any record's count row) pam_frame* does NOT apply the frame --
 * the state stays as it is and the record carries status bit 16.  The host re-submits from record word [3] once it has cleared the word.
 * dev_word = NULL removes the guard. */
int pam_set_input_guard(PamHandle* h, const int32_t* dev_word);

/* ---- per-operator entry points (parity tests; host buffers, synchronous) -----------------------------------*/
/* Camera.projectPoints_parallel, ivclabpose.py:91-98: n poses (17x3) -> (17x2) in (y, x) */
int pam_op_project(PamHandle* h, int cid, int n, const double* poses3d, double* out_yx);
/* affinity build, IterativeTracker.py:137-149: tracks_pose n*17*3, dt n, dets m*17*3 -> aff n*m */
int pam_op_track_affinity(PamHandle* h, int cid, int n, int m, const double* tracks_pose, const int32_t* dt,
                          const double* dets, double* aff);
/* scipy.optimize.linear_sum_assignment call sites IterativeTracker.py:79,150: cost nr*nc (minimise) ->
 * rows/cols (min(nr,nc) pairs, sorted by row) */
int pam_op_lsap(PamHandle* h, int nr, int nc, const double* cost, int32_t* rows, int32_t* cols, int32_t* n_pairs);
/* epipolar_affinity_parallel, matching.py:115-151: V views of one person -> V*V*17 symmetric distances */
int pam_op_epi_dist(PamHandle* h, int V, const int32_t* cids, const double* pose_mat, double* dist);
/* epipolar_distance, matching.py:50-91 (OpenCV epilines form): -> 17*2 */
int pam_op_epi_pair(PamHandle* h, int c1, const double* person1, int c2, const double* person2, double* out);
/* epipolar_affinity, matching.py:93-113 (float32 storage): -> V*V*17 float32 */
int pam_op_epi_dist_init(PamHandle* h, int V, const int32_t* cids, const double* pose_mat, float* dist);
/* Greedy_matching, matching.py:243-295.  mode 0 = 'update' (aff float64, needs pose_j V*3 and next_pose_j 3),
 * mode 1 = 'init' (aff given as float32).  aff is V*V for ONE joint.  -> keep bitmask over views */
int pam_op_greedy(PamHandle* h, int mode, int V, const int32_t* cids, const void* aff, const double* pose_j,
                  const double* next_pose_j, uint32_t* keep_mask);
/* SVD_pose_kernel_jf, construction.py:89-114: keep_mask[17] view sets, Ts[V] ages -> 17*3 */
int pam_op_dlt(PamHandle* h, int V, const int32_t* cids, const int32_t* Ts, const double* pose_mat,
               const uint32_t* keep_mask, const double* next_pose, double* out);
/* IterTrack.smooth_3dpose, IterativeTracker.py:371-383: hist L*17*3 + raw 17*3 -> 17*3 */
int pam_op_smooth(PamHandle* h, int L, const double* hist, const double* raw, double* out);
/* IterTrack.update_motion, IterativeTracker.py:385-395: hist L*17*3 (L>=2) -> float32 17*3 */
int pam_op_velocity(PamHandle* h, int L, const double* hist, float* vel);
/* Hypothesis.calculate_cost, hypothesis.py:53-68 */
int pam_op_hyp_cost(PamHandle* h, int n_members, const int32_t* cids, const double* poses, int o_cid,
                    const double* o_pose, double* cost, int32_t* veto);

/* ---- image side of a1 (HRNetPose.predict pre/post-processing; call site ivclabpose.py:210) ------------------
 * pam_preprocess_crops: n person boxes (x, y, w, h float32, in pixels of frame view_of[i]) cropped from BGR uint8
 * frames (dev pointers, H x W x 3, row pitch W*3), bilinear-resized to out_h x out_w, BGR->RGB, /255, ImageNet
 * mean/std, written as bf16 NHWC (n, out_h, out_w, 3) -- the channels-last input of the conv stack.
 * pam_decode_heatmaps: heat-maps (n, hm_h, hm_w, 17) float32 NHWC or (n,17,hm_h,hm_w) NCHW (nchw flag) -> per person
 * hard arg-max per joint mapped through the box; writes float64 rows (y, x, score) into the tracker's det buffer
 * slot (view_of[i], slot_of[i]).  All pointers are device pointers; asynchronous on `stream`. */
/* final 1x1 convolution of the pose network (`final_layer` of the HRNet module inside HRNetPose.predict): features NHWC bf16
 * (n_pix pixels x C channels, C % 8 == 0) -> heat-maps NHWC float32 (n_pix x J, J = 17); w [J][C] float32, bias [J] or NULL;
 * float32 FMA chain over the channels in order, starting from the bias. */
int pam_head_heatmaps(void* stream, int n_pix, const void* feat_bf16, int C, const float* w, const float* bias, int J,
                      float* out);
/* head + decode in one pass (what the product path runs): the same 1x1 convolution (bit-identical values) with the per-joint
 * (max, first index) reduced on the fly -- the heat-maps are written only if dev_heatmaps_or_null is given.  feat: (n, hm_h, hm_w, C)
 * bf16 NHWC; det / kp / view_of / slot_of / boxes as pam_decode_heatmaps; dev_scratch: pam_head_decode_scratch_bytes(n, hm_h, hm_w)
 * bytes of device memory owned by the caller (per-tile candidates). */
long long pam_head_decode_scratch_bytes(int n, int hm_h, int hm_w);
int pam_head_decode(void* stream, int n, int hm_h, int hm_w, const void* feat_bf16, int C, const float* w, const float* bias, int J,
                    float* dev_heatmaps_or_null, const int32_t* dev_view_of, const int32_t* dev_slot_of, const float* dev_boxes,
                    int max_dets, double* dev_det, float* dev_kp_xyc, void* dev_scratch);
/* the same pass with a SOFT arg-max decode (optional mode; the hard arg-max above is the parity mode): per joint the keypoint is the
 * softmax(beta * heat-map)-weighted mean (column, row) over the crop's heat-map -- sub-pixel -- mapped through the box like the hard
 * one; the confidence stays the heat-map maximum.  Per-tile partials (max, sum exp, weighted sums) merged streaming-softmax style;
 * dev_scratch: pam_head_decode_soft_scratch_bytes(n, hm_h, hm_w) bytes.  beta > 0. */
long long pam_head_decode_soft_scratch_bytes(int n, int hm_h, int hm_w);
int pam_head_decode_soft(void* stream, int n, int hm_h, int hm_w, const void* feat_bf16, int C, const float* w, const float* bias, int J,
                         float beta, float* dev_heatmaps_or_null, const int32_t* dev_view_of, const int32_t* dev_slot_of,
                         const float* dev_boxes, int max_dets, double* dev_det, float* dev_kp_xyc, void* dev_scratch);
int pam_preprocess_crops(void* stream, int n, const void* const* dev_frames /*dev array of n_views frame ptrs*/,
                         int frame_h, int frame_w, const int32_t* dev_view_of, const float* dev_boxes,
                         int out_h, int out_w, int out_c /*3, or 8 = RGB + 5 zero channels*/, void* dev_out_bf16);
/* the same with (a) n_total >= n crops written: crops n .. n_total - 1 repeat crop n - 1 (a hipGraph replay captured for a bucket of
 * n_total crops takes a call of n without a padded box table), and (b) antialias != 0: the resize of upstream simple-HRNet (a PIL image
 * through torchvision's Resize, i.e. PIL's ImagingResample with the bilinear filter): taps = the frame pixels whose centres lie within
 * max(scale, 1) of the output pixel's centre inside the box rounded outwards, triangle weights, normalised, in float32 (PIL's uint8
 * rounding between its two passes is not reproduced).  Equal to the plain bilinear form wherever the box is not larger than the output. */
int pam_preprocess_crops_ex(void* stream, int n, int n_total, const void* const* dev_frames, int frame_h, int frame_w,
                            const int32_t* dev_view_of, const float* dev_boxes, int out_h, int out_w, int out_c, void* dev_out_bf16,
                            int antialias);
int pam_decode_heatmaps(void* stream, int n, const float* dev_heatmaps, int nchw, int hm_h, int hm_w,
                        const int32_t* dev_view_of, const int32_t* dev_slot_of, const float* dev_boxes,
                        int max_dets, double* dev_det, float* dev_kp_xyc /*optional n*17*3 (x,y,conf) or NULL*/);
/* measurement aid, not part of the path: one wave spins `microseconds` and writes {delta shader cycles (s_memtime), delta 100 MHz
 * ticks (s_memrealtime)} to dev_out2 -> shader clock [MHz] = 100 * out[0] / out[1] at that point of the stream (bench.py records it
 * right before and right after the timed region). */
int pam_clock_probe(void* stream, unsigned long long* dev_out2, int microseconds);

/* ---- HRNet conv stack (a1) as hand-written MFMA kernels -------------------------------------------------------
 * pam_conv2d_nhwc_bf16: NHWC bf16 convolution (KH,KW in {1,3}; stride 1/2; Cin % 8 == 0; Cout % 48 == 0 or % 64 == 0) as an
 * implicit GEMM on v_mfma_f32_16x16x32_bf16 with fused epilogue out = act(conv + bias [+ residual]); `relu` is the activation
 * code: 0 linear, 1 ReLU, 2 leaky ReLU (slope 0.1, Darknet); + 4 = add the residual AFTER the activation (Darknet shortcut).  w_packed is
 * [Cout][Kpad] bf16, k = (ky, kx, cin) flattened, zero-padded to Kpad = roundup(KH*KW*Cin, 64); bias float32 or NULL;
 * residual NHWC bf16 of the output shape or NULL.  tile_cfg: -1 = choose automatically (w_img in the layout pam_conv3x3_layout() announces
 * at call time), -3 / -4 = the same with the layout of w_img STATED by the caller (-3 streamed: k_conv3x3s or PAM_E_ARG; -4 classic),
 * -2 = the classic kernels (k_conv3x3 /
 * k_conv_igemm) with automatic tiles, 0..7 = one k_conv_igemm tile shape, 8 = the streamed implicit GEMM k_conv_gs (activation codes 0 / 1).  w_img (optional, 3x3 stride-1 layers
 * with Cin in {48,64,96,128,192,256,384,512}): the same weights pre-packed as per-chunk LDS images [Cout/BN][Cin/CK][BN][9*CK + pad] (BN =
 * pam_conv3x3_slab(H, W, Cin, Cout); CK = 48 if Cin == 48, 64 if Cin >= 192, else 32; row pitch 864 / 1184 / 608 bytes;
 * row j*16 + q of a slab, q < 16, j < BN/16, holds output channel 4*(BN/16)*(q >> 2) + 4*j + (q & 3) of that slab) for the
 * rows-in-LDS kernel k_conv3x3; for a stem layer (Cin 8, Cout 32 or 64, 3x3, stride 1 or 2, pad 1; Cout 32 is accepted only
 * here) w_img is instead the MFMA A fragments of k_conv_stem, [j < Cout/16][ky < 3][lane < 64][8] bf16: lane l = input channels
 * 0..7 of tap (ky, kx = l >> 4; zero for l >= 48) of output channel (Cout/4)*((l & 15) >> 2) + 4*j + (l & 3).  NULL = generic kernel.
 * pam_upsample_add_nhwc_bf16: the HRNet fuse-layer sum out = [relu](base + sum_t nearest_upsample(term_t, 2^shift_t)). */
int pam_conv2d_nhwc_bf16(void* stream, const void* in, const void* w_packed, const void* w_img, const float* bias,
                         const void* residual, void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, int relu, int tile_cfg);
/* the same with channel-sliced operands, for the merged fuse-layer convolutions: the input pixels are in_cstride channels apart
 * (`in` points at the first of the Cin channels read; 0 = Cin), and the activation applies only to output channels >= relu_from
 * (multiple of 16; 0 = all).  Either option selects the generic kernel. */
int pam_conv2d_nhwc_bf16_ex(void* stream, const void* in, const void* w_packed, const void* w_img, const float* bias,
                            const void* residual, void* out, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                            int stride, int pad, int relu, int tile_cfg, int in_cstride, int relu_from);
/* output channels per workgroup slab (BN) that k_conv3x3 uses for a layer shape; weight images must be packed with it */
int pam_conv3x3_slab(int H, int W, int Cin, int Cout);
/* 0: w_img of this layer is the classic per-chunk image described above; BN > 0: the layer runs on the streamed kernel k_conv3x3s
 * (Cin 192 / 384 at tile_cfg == -1) with slabs of BN output channels and w_img must be [Cout/BN][Cin/32][9 taps][BN rows][4][8]
 * bf16: chunk c, tap t, row r (same row -> channel rule as above) holds input channels 32*c + 8*g .. + 8 of that tap in its 16-byte
 * piece g ^ ((r >> 1) & 2).  tile_cfg == -2 forces the classic kernel (and the classic image) for such a layer.  BN > 0 is only
 * returned for shapes the streamed kernel is instantiated for (64-channel slabs for Cin 64 / 192 / 384, the 48-channel slab for
 * 256 -> 48); the streamed kernel takes activation codes 0 / 1 only, so a layer with a Darknet code (relu > 1) must be packed in the
 * classic layout and called with tile_cfg == -2 (tile_cfg == -1 with a streamed image and relu > 1 returns PAM_E_ARG). */
int pam_conv3x3_layout(int H, int W, int Cin, int Cout);
/* the same with the 96 -> 96 layers' choice stated: c96_slab 0 = they stay on k_conv3x3 (the answer above), 48 = streamed with slabs
 * of 48 output channels (the only streamed slab width instantiated for these layers; any other value answers as 0).  The caller packs
 * the image for the slab returned and says so at the launch: tile_cfg -5 of pam_conv2d_nhwc_bf16_ex, which otherwise behaves like -3
 * (streamed layout stated; -4 = classic layout stated). */
int pam_conv3x3_layout_ex(int H, int W, int Cin, int Cout, int c96_slab);
/* Darknet's 3x3 stride-1 layers (activation codes > 1: leaky, shortcut added after the activation) on the streamed kernel: the slab
 * width (64 for Cin 128, 32 for Cin 256 / 512) when the library has a general-activation instantiation for the shape (Cout % 64 == 0,
 * rows that fit the patch), else 0.  The caller then packs the streamed image for slabs of that width and launches with tile_cfg -7 (any other tile_cfg keeps such layers on
 * the classic kernel and image). */
int pam_conv3x3_layout_gen(int H, int W, int Cin, int Cout);
/* round 5: 192- / 384-channel ReLU / linear 3x3 layers with 32-channel slabs (tile_cfg -8 of pam_conv2d_nhwc_bf16_ex; w_img packed as
 * above for slabs of 32): twice the workgroups, each half as long -- for forwards of a few crops, whose launches are as long as one
 * workgroup.  Same arithmetic and summation order as the 64-channel-slab form (bit-identical).  > 0: supported, the slab width (32). */
int pam_conv3x3_layout_small(int H, int W, int Cin, int Cout);
/* which kernel the calling thread's last pam_conv2d_nhwc_bf16[_ex] call launched (labels for per-kernel profiles) */
#define PAM_CONV_KERNEL_IGEMM 0   /* k_conv_igemm */
#define PAM_CONV_KERNEL_3X3   1   /* k_conv3x3   */
#define PAM_CONV_KERNEL_3X3S  2   /* k_conv3x3s  */
#define PAM_CONV_KERNEL_GS    3   /* k_conv_gs   */
#define PAM_CONV_KERNEL_STEM  4   /* k_conv_stem */
int pam_conv_last_kernel(void);
/* diagnostic builds only: device buffer (64 x uint64 per workgroup) for k_conv3x3's s_memtime stamps, used when tile_cfg = 100 + 64 */
int pam_conv_debug_stamps(void* dev_buf);
int pam_upsample_add_nhwc_bf16(void* stream, const void* base, int n_terms, const void* const* terms,
                               const int32_t* shifts, void* out, int N, int H, int W, int C, int relu);
/* the same with terms that are channel slices of wider tensors: term_cstrides[t] = channels between pixels of term t (NULL / 0 = C) */
int pam_upsample_add_nhwc_bf16_ex(void* stream, const void* base, int n_terms, const void* const* terms, const int32_t* shifts,
                                  const int32_t* term_cstrides, void* out, int N, int H, int W, int C, int relu);

/* ---- fused HRNet BasicBlock (row a1; stands inside the absent HRNet backend behind /root/reference/src/ivclabpose.py:210).
 * out = ReLU(conv3x3(ReLU(conv3x3(in) + b1)) + b2 + in), NHWC bf16, C -> C -> C, stride 1, pad 1, BatchNorm folded; the intermediate
 * never leaves LDS.  ONE launch per tensor with 2-D items of tile_rows x tile_cols output positions (csrc/pam_block2.hip), all
 * operands fetched by LDS-DMA.  pam_basic_block2_tile writes the {rows, cols} the library would pick for N x H x W (whole rounds of 256
 * workgroups first, then the least work per item); tile_rows / tile_cols <= 0 in the launch = that choice.  in != out.
 * C = 48 (k_bblock2_48): BOTH weight sets resident in LDS beside a 75 KB input tile, no barrier inside the K loops.  Limits:
 *   (rows + 4)(cols + 4) <= 800, (rows + 2)(cols + 4) <= 768, rows (cols + 2) <= 640.
 *   wpack (1 024 + 86 016 bytes): [float32 bias: conv1's 48, conv2's 48, zero padding to 1 KiB][conv1's 14 k-step images][conv2's 14]; a
 *   k-step = 32 K elements, K = flattened (tap, cin) index padded with zeros to 14*32; a k-step image = [48 rows][64 bytes = 4 pieces
 *   of 8 bf16]: row j*16 + q = output channel 8*(q >> 2) + 4*j + (q & 3) for j < 2 and 32 + 4*(q >> 2) + (q & 3) for j = 2, and
 *   physical piece p of row R holds K elements 8*(p ^ s) .. + 7 of the k-step with s = (0,2,3,1)[(R%16) >> 2] (LDS bank swizzle).
 *   Results are bit-identical to two pam_conv2d_nhwc_bf16 calls (same K order per output element).
 * C = 96 (k_bblock2_96<3,2>): the input tile resident (chunk-major, 3 x 32 channels), the weights of both convolutions
 *   streamed through a ring of six k-step images (twelve while the tile's (rows + 4)(cols + 4) slots, rounded up to 16, stay <= 448).
 *   Limits of the one instantiation shipped (<3,2>): (rows + 4)(cols + 4) <= 640, (rows + 2)(cols + 4) <= 384, rows (cols + 2) <= 256.
 *   wpack (1 024 + 331 776 bytes): [float32 bias 96 + 96, padded to 1 KiB][54 k-step images of [96 rows][64 bytes]] in the order (conv,
 *   chunk of 32 input channels, tap); row j*16 + q = output channel 24*(q >> 2) + 4*j + (q & 3); physical 16-byte piece p of row r
 *   holds the chunk's input channels 8*(p ^ ((r >> 1) & 2)) .. + 7.  The residual enters the sum before the products (as in the
 *   streamed convolution kernel): equal to the two-launch path to within the last bf16 bit. */
int pam_basic_block2_tile(int C, int N, int H, int W, int32_t* out2);
int pam_basic_block2_nhwc_bf16(void* stream, const void* in, const void* wpack, void* out, int N, int H, int W, int C,
                               int tile_rows, int tile_cols);

/* ---- the pointwise tail of a layer1 Bottleneck as ONE launch (row a1; csrc/pam_pw.hip) ------------------------------------------
 * X = ReLU(W3 . y2 [+ Wd . x0] + bias3 [+ residual]);  y1 = ReLU(W1 . X + bias1)   per pixel, NHWC bf16:
 * y2 (n_pixels x 64): the block's 3x3 output; x0 (n_pixels x 64, or NULL): the first block's input, whose 1x1 downsample convolution is
 * a second K range of the same product (bias3 then = conv3's + the downsample's); residual (n_pixels x 256, or NULL; not with x0);
 * out_x (n_pixels x 256); w1_img / bias1 / out_y1 (n_pixels x 64): the NEXT block's conv1, or all NULL.  HRNet-W48's widths only.
 * w3_img [S][256][64] bf16 (S = 1, or 2 with x0): chunk c = K source c; row 64 sl + 16 jt + qq (qq < 16) = output channel
 *   64 sl + 16 (qq >> 2) + 4 jt + (qq & 3); the row's 16-byte piece at position p holds K values 8 q .. 8 q + 7, q = p ^ ((row >> 1) & 7).
 * w1_img [4][64][64] bf16: chunk sl = input channels 64 sl .. + 63; row 16 jt + qq = output channel 16 (qq >> 2) + 4 jt + (qq & 3);
 *   piece p holds, with q = p ^ ((row >> 1) & 7), h = q >> 2, g = q & 3, the input channels 64 sl + 16 g + 8 h .. + 7.
 * tile_cfg: 16-pixel tiles per wave tile (1..3), <= 0 = automatic.  replaces: conv3 + bn3 + residual + relu and the next conv1 + bn1 +
 * relu of the official Bottleneck (the HRNet backend is absent from the reference: call sites /root/reference/src/ivclabpose.py:131-132,210). */
/* Round 4: HRNet's stem and the first Bottleneck's conv1 as ONE launch (csrc/pam_stem.hip):
 *   c1 = ReLU(conv3x3 s2 p1 (in; 8 -> 64) + bias1), x0 = ReLU(conv3x3 s2 p1 (c1; 64 -> 64) + bias2), y1 = ReLU(conv1x1 (x0; 64 -> 64) + biasp);
 * c1 stays in LDS.  in (N, H, W, 8) bf16 NHWC, out_x0 / out_y1 (N, H2, W2, 64) with H1 = (H - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1 (W alike).
 * w1frag: the 8 -> 64 layer's w_img of pam_conv2d_nhwc_bf16 (k_conv_stem's A fragments, [4][3][64 lanes][8]); wp_img / biasp:
 * pam_pointwise64_relu_nhwc_bf16's; w2img [9 taps][64 rows][64 K] bf16: row 16 j + q of a tap = output channel
 * 32 (j >> 1) + 8 (q >> 2) + 4 (j & 1) + (q & 3), the row's 16-byte piece at position p holds input channels 8 c .. 8 c + 7 with
 * c = p ^ ((q >> 1) & 7).  Results are bit-identical to the three launches it replaces (pam_conv2d_nhwc_bf16 twice, then
 * pam_pointwise64_relu_nhwc_bf16).  replaces: conv1/bn1/relu, conv2/bn2/relu and layer1[0].conv1/bn1/relu of the official pose_hrnet (the
 * HRNet backend is absent from the reference: call sites /root/reference/src/ivclabpose.py:131-132,210). */
int pam_stem_fused_nhwc_bf16(void* stream, const void* in, const void* w1frag, const float* bias1, const void* w2img, const float* bias2,
                             const void* wp_img, const float* biasp, void* out_x0, void* out_y1, int N, int H, int W);
/* Round 4: the 3x3 convolution of a layer1 Bottleneck and its pointwise tail as ONE launch (csrc/pam_bneck.hip):
 *   y2 = ReLU(conv3x3 s1 p1 (y1; 64 -> 64) + bias2), then pam_bottleneck_tail_nhwc_bf16's X = ReLU(W3 . y2 + bias3 + residual) and
 *   y1' = ReLU(W1 . X + bias1) (w1_img / bias1 / out_y1 all NULL: no second product); y2 never leaves the registers.
 * y1 (N, H, W, 64), residual / out_x (N, H, W, 256), out_y1 (N, H, W, 64), bf16 NHWC.  Exactly one of residual and x0 (N, H, W, 64: the
 * FIRST block's input, whose 1x1 downsample convolution is the second K source of w3_img, as in pam_bottleneck_tail_nhwc_bf16) is given.
 * w2img: the [9 taps][64 rows][64 K] image of pam_stem_fused_nhwc_bf16; w3_img (one K source, two with x0) / w1_img:
 * pam_bottleneck_tail_nhwc_bf16's.  Results are bit-identical to
 * pam_conv2d_nhwc_bf16 (streamed 3x3 kernel) followed by pam_bottleneck_tail_nhwc_bf16.  replaces: conv2/bn2/relu, conv3/bn3 + residual +
 * relu and the next conv1/bn1/relu of the official Bottleneck (call sites /root/reference/src/ivclabpose.py:131-132,210). */
int pam_bottleneck_fused_nhwc_bf16(void* stream, const void* y1, const void* x0, const void* residual, const void* w2img,
                                   const float* bias2, const void* w3_img, const float* bias3, const void* w1_img, const float* bias1,
                                   void* out_x, void* out_y1, int N, int H, int W);
/* y = ReLU(W . x + bias), 64 -> 64 channels, pointwise (the first Bottleneck's conv1 on the stem output): w_img [64 rows][64 K] bf16, row
 * 16 jt + qq = output channel 16 (qq >> 2) + 4 jt + (qq & 3), natural K order, the row's 16-byte piece at position p holds K values
 * 8 q .. 8 q + 7 with q = p ^ ((row >> 1) & 7). */
int pam_pointwise64_relu_nhwc_bf16(void* stream, const void* in, const void* w_img, const float* bias, void* out, long long n_pixels);
/* the same stream with the activation stated: act 1 = ReLU, 2 = leaky ReLU of slope 0.1 (Darknet's 64 -> 32 pointwise layer, its 32 filters
 * zero-padded to 64, at 208 x 208) */
int pam_pointwise64_act_nhwc_bf16(void* stream, const void* in, const void* w_img, const float* bias, void* out, long long n_pixels, int act);
int pam_bottleneck_tail_nhwc_bf16(void* stream, const void* y2, const void* x0, const void* residual, const void* w3_img,
                                  const float* bias3, const void* w1_img, const float* bias1, void* out_x, void* out_y1,
                                  long long n_pixels, int tile_cfg);

/* ---- round 5: the fuse layers of an HR module (row a1; hrnet.py:64-100 -- stands inside the absent HRNet backend behind
 * /root/reference/src/ivclabpose.py:210): every output i of a module is ReLU(x_i + strided-convolution chains from the finer branches +
 * up-sampled 1x1 convolutions of the coarser ones).
 *
 * pam_conv3x3s2_c48_nhwc_bf16 (csrc/pam_down.hip, k_down48): out = [ReLU on channels >= relu_from](conv3x3 stride 2 pad 1 (in) + bias
 * [+ res]) for Cin = 48: the input patch of a tile of output positions resident in LDS (columns stored by parity), the output channels
 * walked in 48-channel slabs whose weights are resident and double-buffered.  in: (N, H, W) pixels of in_cstride channels (the first 48
 * are read: a channel slice of a wider tensor is fine); out / res: (N, Ho, Wo) pixels of out_cstride / res_cstride channels, Ho = (H - 1)
 * / 2 + 1; Cout % 48 == 0.  wpack: Cout / 48 slab images of 43 008 bytes, slab s = output channels 48 s .. + 47 in the layout of ONE
 * convolution of pam_basic_block2_nhwc_bf16's C = 48 image ([14 k-steps][48 rows][64 bytes], see there).  tile_rows / tile_cols <= 0 and
 * slab_groups <= 0: the library's choice (pam_conv3x3s2_c48_tile writes {rows, cols, groups}); slab_groups = workgroups an item's slabs
 * are spread over (divides Cout / 48).  Results are bit-identical to pam_conv2d_nhwc_bf16 (same K order per output element).
 *
 * pam_fuse_sum_nhwc_bf16 (csrc/pam_fuse.hip, k_fuse_sum): out = [ReLU](base + sum_t plain_t + sum_s nearest_up_{2^shift_s}(W_s . src_s +
 * bias_s)): base / out / plain_t (N, H, W, C) NHWC bf16, C in {48, 96, 192} (plain_t may be channel slices: plain_cstrides, NULL = C);
 * src_s (N, H >> shift_s, W >> shift_s, up_channels[s]) with ascending shifts, up_channels % 32 == 0, H and W multiples of 2^shift;
 * up_wimg[s]: the 1x1 weights (C x up_channels[s]) as MFMA A fragments, bf16 [C / 16][up_channels / 32][64 lanes][8]: lane l of fragment
 * (j, ks) holds W[16 j + (l & 15)][32 ks + 8 (l >> 4) .. + 7]; up_bias[s]: float32 [C] or NULL.  n_plain <= 2, 1 <= n_up <= 3.  Persistent
 * workgroups (at most one per CU, at most max_workgroups when that is > 0: several outputs of a module then share the chip) keep all the
 * 1x1 weights in LDS and walk tiles of tile_a x tile_b pixels of the coarsest source (<= 0: the library's choice, 1 x 3); the products
 * never reach HBM; sum order = base, plain terms, up terms (the order of pam_upsample_add_nhwc_bf16 with the terms in branch order).  Results are
 * bit-identical to pam_conv2d_nhwc_bf16 (1x1) per source followed by pam_upsample_add_nhwc_bf16. */
int pam_conv3x3s2_c48_tile(int N, int H, int W, int Cout, int32_t* out3);
/* the same convolution for Cin = 96 / 192 (k_down_s: the streamed 3x3 kernel with a stride-2, parity-planar patch; no residual).
 * pam_conv3x3s2_slab: the slab width BN (64 or 48 output channels) the library takes for this shape, 0 = not taken (generic kernel);
 * w_img: the streamed 3x3 image of pam_conv2d_nhwc_bf16 for that BN ([Cout / BN][Cin / 32][9 taps][BN rows][4 pieces of 8 bf16]: row
 * j*16 + q of a slab = its channel 4*(BN/16)*(q >> 2) + 4*j + (q & 3), physical piece p of row r = the chunk's input channels
 * 8*(p ^ ((r >> 1) & 2)) .. + 7).  relu_from % 16 == 0.  Equal to pam_conv2d_nhwc_bf16 up to the summation order (K walked chunk by
 * chunk of 32 input channels instead of tap by tap): within one bf16 rounding of the fp32 result. */
int pam_conv3x3s2_slab(int H, int W, int Cin, int Cout);
int pam_conv3x3s2_nhwc_bf16(void* stream, const void* in, int in_cstride, const void* w_img, const float* bias, void* out, int N, int H,
                            int W, int Cin, int Cout, int relu, int relu_from);
int pam_conv3x3s2_c48_nhwc_bf16(void* stream, const void* in, int in_cstride, const void* wpack, const float* bias, const void* res,
                                int res_cstride, void* out, int out_cstride, int N, int H, int W, int Cout, int relu, int relu_from,
                                int tile_rows, int tile_cols, int slab_groups);
int pam_fuse_sum_nhwc_bf16(void* stream, const void* base, int n_plain, const void* const* plain, const int32_t* plain_cstrides, int n_up,
                           const void* const* up_src, const int32_t* up_shifts, const int32_t* up_channels, const void* const* up_wimg,
                           const float* const* up_bias, void* out, int N, int H, int W, int C, int relu, int tile_a, int tile_b,
                           int max_workgroups);

/* ---- round 5: device-side ordering of the forward's branch streams (csrc/pam_sync.hip; the module-end exchange of hrnet.py:64-100 inside
 * the absent HRNet backend, /root/reference/src/ivclabpose.py:210).  pam_flag_signal: one agent-scope atomic add on *dev_counter behind
 * everything already queued on `stream`.  pam_flag_gate: [arrive != 0: first the same add, then] `stream` goes on once *dev_counter >= target (one wave polls; the launches that
 * follow on the stream see what the signalling streams' earlier kernels wrote); after *dev_max_us microseconds (a device word read when
 * the gate starts: the bound of a captured gate can be changed between replays) it sets *dev_err = 1 and lets the stream go on
 * regardless -- the caller checks dev_err (and host_err, when given: a word of pinned host memory that receives 1 at a time-out, readable
 * by the host without touching the device; and dev_void, when given: a device word that receives 1, for consumers of the forward's
 * results on the device -- pam_set_input_guard; NULL = none).  Counters are zeroed by the caller in front of the first signal.  Flagged forwards of one process must not be in flight at the same time (their gates can block each other's queues). */
int pam_flag_signal(void* stream, int32_t* dev_counter);
/* the same add on every counter dev_counters[k] whose bit k of mask is set (k < 32), one launch, one release */
int pam_flag_signal_mask(void* stream, int32_t* dev_counters, uint32_t mask);
int pam_flag_gate(void* stream, int32_t* dev_counter, int target, int32_t* dev_err, const uint32_t* dev_max_us, int arrive, int32_t* host_err,
                  int32_t* dev_void);

/* ---- row e: the path's one exchange, in the C ABI (the reference has no distributed code -- it hands every visible GPU
 * to HRNet, /root/reference/src/ivclabpose.py:107-111,131-132).  One process per GPU; camera views are partitioned over the ranks; before
 * the cross-view match every rank contributes its views' keypoint records and receives everyone's: ONE all-gather per frame, enqueued on
 * the stream the decode ran on.  Record per view: (max_dets + 1) x 17*3 float64 -- max_dets detection rows (y, x, score), i.e. exactly
 * what pam_head_decode writes for that view when it is given max_dets + 1 as its slot stride, then one more row whose FIRST double is
 * the view's detection count; rows_per_rank = the largest number of views any rank owns (unused records are padding); dev_recv holds
 * world * rows_per_rank records, rank-major, and is what pam_frame_dev_views reads.
 * The communicator is RCCL's (ncclComm_t); librccl is bound at run time, so single-GPU hosts never need it.  pam_comm_unique_id on one
 * rank -> ship the 128 bytes to the others by any means -> pam_comm_init on every rank (collective). */
int pam_comm_unique_id(void* id128);
int pam_comm_init(void** comm, int world, int rank, const void* id128, int device);
int pam_comm_destroy(void* comm);
const char* pam_comm_last_error(void);
int pam_allgather_keypoints(PamHandle* h, void* comm, void* stream, const double* dev_send, int rows_per_rank, double* dev_recv);

/* ---- person detector side (ivclabpose.py:116-120 constructs backend.YOLOv3, :183-204 PersonDetect calls it).
 * The backend is absent from the reference tree; these follow the public Darknet YOLOv3 definition (parity unpinned).
 * pam_resize_frames: n BGR uint8 frames (dev array of dev pointers, H x W x 3) -> bilinear (cv2.resize INTER_LINEAR
 * semantics) out_h x out_w, BGR->RGB, /255, bf16 NHWC with 8 channels (RGB + 5 zeros).
 * pam_upsample_concat_nhwc_bf16: Darknet `upsample` + `route`: out[n,y,x] = concat(a[n,y/2,x/2,:Ca], b[n,y,x,:Cb]).
 * pam_yolo_detect: the three `yolo` heads (NHWC bf16, anchor-major channels [tx,ty,tw,th,obj,cls...] x 3, channel
 * stride chan_stride[h] >= 3*(5+num_classes)) -> per image the boxes of class `class_id` with
 * sigmoid(obj)*sigmoid(cls) > score_thresh, greedy NMS (IoU > nms_thresh suppressed, best score first, ties to the
 * lower candidate number: head, then cell row-major, then anchor), at most max_det rows (x1, y1, x2, y2, score) in
 * frame pixels (float32) into dev_out[n_img][max_det][5]; dev_count[i] = rows kept, dev_count[n_img + i] = boxes above
 * the threshold before NMS (only the first PAM_YOLO_MAX_CAND of them, in candidate order, enter NMS).
 * anchors: [head][anchor][w, h] in network-input pixels (18 floats, host memory). */
int pam_resize_frames(void* stream, int n, const void* const* dev_frames, int frame_h, int frame_w,
                      int out_h, int out_w, void* dev_out_bf16);
int pam_upsample_concat_nhwc_bf16(void* stream, const void* a, const void* b, void* out, int N, int H, int W, int Ca, int Cb);
int pam_yolo_detect(void* stream, int n_img, const void* const* heads /*host array of 3 dev ptrs*/, const int32_t* grid_h,
                    const int32_t* grid_w, const int32_t* chan_stride, const float* anchors, int net_w, int net_h,
                    int num_classes, int class_id, float score_thresh, float nms_thresh, int frame_w, int frame_h,
                    int max_det, float* dev_out, int32_t* dev_count);
/* pam_yolo_detect_ws (round 5): the same result with the scoring pass spread over several workgroups per image (2 048 candidates each;
 * the workgroup that finishes an image last concatenates their kept boxes in candidate order and runs the NMS).  dev_workspace: at least
 * pam_yolo_detect_workspace_bytes(n_img, grid_h, grid_w) bytes of device memory, 16-byte aligned, whose first 4 * n_img bytes are ZERO
 * before the first call (the kernel leaves them zero); one workspace serves one call at a time. */
long long pam_yolo_detect_workspace_bytes(int n_img, const int32_t* grid_h, const int32_t* grid_w);
int pam_yolo_detect_ws(void* stream, int n_img, const void* const* heads, const int32_t* grid_h, const int32_t* grid_w,
                       const int32_t* chan_stride, const float* anchors, int net_w, int net_h, int num_classes, int class_id,
                       float score_thresh, float nms_thresh, int frame_w, int frame_h, int max_det, float* dev_out, int32_t* dev_count,
                       void* dev_workspace, long long workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PAM_H */
