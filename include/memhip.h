/*
 * memhip.h -- C ABI of libmemhip.so: the MI355X (gfx950) kernels behind the
 * MEM (tum-vision/mem) pretraining hot path.
 *
 * The reference has no FFI of its own (it is pure Python):
 * the drop-in boundary is its Python surface, mirrored in the mem_amd package, and
 * THIS header is what that mirror binds with ctypes.  Each entry point cites
 * the reference code it replaces (paths relative to /root/reference).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes only; pointers are caller-owned DEVICE buffers
 *     (tensor.data_ptr()) unless a parameter is documented "host".
 *   - no allocation, no host synchronisation, no global mutable state inside
 *     that the product path writes; scratch comes in through (workspace,
 *     workspace_bytes); work is enqueued on `stream` (a hipStream_t passed as
 *     void*; NULL = default stream).  The two exceptions, both explicit calls:
 *     memhip_set_option (a process-wide kernel-selection table for A/B
 *     measurements; the mem_amd package never calls it outside tools and
 *     tests) and memhip_stream_reserve_cus (a reservation keyed by the
 *     caller's stream handle, set and cleared by the data-parallel reducer).
 *   - return 0 on success, a negative MEMHIP_E* code on failure; a message is
 *     available from memhip_last_error() (thread-local).  No C++ exception
 *     crosses the boundary.
 *   - "rows_pad": token-major activation matrices are allocated with their row
 *     count rounded up to a multiple of 128 and zero-filled once; kernels never
 *     write rows >= rows.
 */
#ifndef MEMHIP_H
#define MEMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEMHIP_ABI_VERSION 5   /* 5 (round 6): memhip_build_flags, memhip_attn_bwd_ws / _out_ws / _workspace; 4 (round 5): epilogues 6 / 7 carry the stored GELU derivative as FP16 (since round 4), certified-tokenizer entry points */

#define MEMHIP_OK 0
#define MEMHIP_EINVAL (-1)   /* bad argument (shape / alignment / null) */
#define MEMHIP_EWORKSPACE (-2) /* workspace too small */
#define MEMHIP_ELAUNCH (-3)  /* hip launch / runtime error */
#define MEMHIP_EUNSUPPORTED (-4)

typedef void* memhip_stream_t;

int memhip_abi_version(void);
const char* memhip_last_error(void);
/* Name of the device code object this library was built for ("gfx950"). */
const char* memhip_arch(void);
/* Extra compiler flags this library was built with ("" for the shipped build; a measurement build made by
 * tools/build_variant.sh names its -D switches here, and bench.py prints them next to the path of the library it measured). */
const char* memhip_build_flags(void);
/* Kernel-selection switches for A/B measurements (tools/): the library reads NO environment variable; the only
 * process-wide state is this explicit table.  Names: "gemm_p8", "gemm256", "gemm_split", "gemm_p8_half", "tn_p8",
 * "raster_lds", "attn16" (0/1, default 1 = shipped dispatch), "gemm_p8_min_n" (768), "gemm256_min_n" (1024),
 * "attn16_stagger" (40000) / "attn16_stagger_fwd" (0): cycles by which the 14x14 attention workgroups with the smaller share
 * of samples start late at most, "gemm_stagger" (0): the same for the persistent GEMM workgroups, cycles per K-tile;
 * round 5: "attn_win" (1: windows 40 / 20 tokens wide and longer than 256 tokens run on the slot-layout kernels of
 * attn_win.hip, 0: attn_stream.hip), "gemm_p8_pair" (1: the full rounds and the ragged round of an NT product are ONE
 * launch, 0: two launches), "tn_group" (1: memhip_gemm_bf16_tn_group runs its products as one grid, 0: one by one);
 * round 6: "conv_waves" (16: the fp16x2 tokenizer convolutions run 8 waves per workgroup and the phase-interleaved 256 x 128 tile
 * on every layer whose grid fills the chip twice; 32: that tile at every size; 8: 8 waves, 128 x 128 tiles only; 4: 4 waves, one per SIMD),
 * "raster_bands" (0: memhip_rasterize_binned_f64 chooses the bands per sample from the batch size; n > 0: n bands, raised to
 * the fewest the canvas allows).
 * Unknown name: MEMHIP_EINVAL.  Results do not depend on any option (same contract, different kernel or timing). */
int memhip_set_option(const char* name, int value);
int memhip_get_option(const char* name, int* value);
/* CUs that launches on `stream` leave free (rounded up to a multiple of 8: every XCD gives up the same number; 0 clears
 * it).  The persistent one-workgroup-per-CU launches (GEMMs, weight gradients, attention) size their grids for the device's
 * CUs minus this: the data-parallel reducer sets it on ITS engine's streams while gradient buckets are in flight, so that
 * RCCL's channel kernels find CUs of their own (mem_amd/parallel.py; DDP: mem/run_mem_pretraining.py:365-367).  Scoped to
 * the caller's stream handle -- another engine / stream of the process is not affected; at most 32 streams at a time. */
int memhip_stream_reserve_cus(memhip_stream_t stream, int cus);

/* ------------------------------------------------------------------------
 * Event stream -> voxel/histogram image
 * replaces EventArrToImg.__call__            mem/datasets.py:566-595
 * ------------------------------------------------------------------------
 * ev        f64 [n_total, 4] rows [x, y, t, p] (the reference's (N,4) float64
 *           layout, mem/dataset_folder.py:275-302), samples concatenated
 * offsets   i64 [B+1] CSR row offsets of each sample inside ev
 * out       u8  [B, 3, H, W] planar [pos, tss, neg] (the reference returns the
 *           same bytes viewed as (H, W, 3); the Python mirror transposes)
 * status    i32 [B]  out: number of events of sample b whose flat pixel index
 *           x + W*y fell outside [-H*W, H*W) (the reference raises IndexError)
 * workspace u32 [B, 3, H*W] (memhip_rasterize_workspace)
 * Semantics: x,y truncated toward zero; only p == +1 / p == -1 count; counts
 * are exact mod 256; NumPy negative-index wrap kept; time surface (optional)
 * = (t - tmin) / (tmax - tmin) * 255 truncated to u8, last event in array
 * order wins.
 */
size_t memhip_rasterize_workspace(int B, int H, int W);
int memhip_rasterize_f64(const double* ev, const int64_t* offsets, int B, int H, int W,
                         int time_surface, uint8_t* out, int32_t* status,
                         void* workspace, size_t workspace_bytes, memhip_stream_t stream);

/* Event-level augmentations fused into the rasterizer's single read of the
 * events (no intermediate (N',4) arrays):
 * replaces ReshapeScaleXandY  mem/datasets.py:464-485   x*=scale_x, y*=scale_y
 *          RandomTimeFlip     mem/datasets.py:598-609   reverse order, t<-t_last-t, p<- -p
 *          Aug_FlipEvsAlongX  mem/datasets.py:501-521   x <- flip_w-1-x
 *          Aug_RandomShiftEvs mem/datasets.py:524-549   x+=shift_x, y+=shift_y, keep
 *                                                       0<=x<filt_w and 0<=y<filt_h
 * applied per event in exactly that order, in float64 like the reference.
 * (SliceRandomMaxEvs, datasets.py:488-498, is a contiguous window: express it
 * through `offsets`.)  The random draws stay with the caller (host), so parity
 * is draw for draw.  aug = device array of B records, or NULL for none.
 */
typedef struct memhip_event_aug {
  double scale_x, scale_y;   /* 1.0 = off */
  int32_t time_flip;         /* 0/1 */
  int32_t flip_x;            /* 0/1 */
  int64_t flip_w;            /* W used by the x flip */
  int32_t shift_x, shift_y;
  int32_t do_filter;         /* 0/1: apply the bounds filter of Aug_RandomShiftEvs */
  int32_t filt_w, filt_h;
  int32_t infer;             /* MEMHIP_AUG_INFER_* bits: fields that memhip_aug_resolve fills from the data (0 = none) */
} memhip_event_aug_t;
#define MEMHIP_AUG_INFER_FLIP_W 1
#define MEMHIP_AUG_INFER_FILT_W 2
#define MEMHIP_AUG_INFER_FILT_H 4
int memhip_rasterize_aug_f64(const double* ev, const int64_t* offsets, const memhip_event_aug_t* aug,
                             int B, int H, int W, int time_surface, uint8_t* out, int32_t* status,
                             void* workspace, size_t workspace_bytes, memhip_stream_t stream);

/* The same rasterizer for LONG streams (N-ImageNet scale, ~1 M events per sample;
 * BASELINE configs[3]) without time surface: two streaming passes (events ->
 * 2-byte pixel keys sorted by band of the canvas -> LDS counters per band)
 * instead of one global atomic per event.  Same semantics, outputs and status
 * as memhip_rasterize_aug_f64(time_surface = 0).
 * n_events  upper bound on offsets[B] - offsets[0] (sizes the key workspace; a
 *           sample that would exceed it gets status |= 1<<30 and a zero image)
 * status    written for every sample by the second pass (the caller zeroes nothing)
 * Canvas limit: H*W <= 64 * 40000 pixels (MEMHIP_EUNSUPPORTED beyond). */
size_t memhip_rasterize_binned_workspace(int B, int H, int W, int64_t n_events);
int memhip_rasterize_binned_f64(const double* ev, const int64_t* offsets, const memhip_event_aug_t* aug,
                                int B, int H, int W, int64_t n_events, uint8_t* out, int32_t* status,
                                void* workspace, size_t workspace_bytes, memhip_stream_t stream);

/* Data-dependent canvases without a host round trip (mem/datasets.py:516,537-540,572-575: "W = x.max() + 1"):
 *   memhip_events_extent(raw window) -> memhip_aug_resolve(stage 0) fills the inferred flip / filter sizes of aug[b];
 *   memhip_events_extent(with aug)   -> memhip_aug_resolve(stage 1) writes the canvas dims[b] = (H_b, W_b)
 *   (fixed_h / fixed_w > 0 override; empty samples or canvases beyond hmax x wmax: dims (0,0), status |= 1 << 29);
 *   memhip_rasterize_var_f64 rasterizes sample b on its own canvas, stored densely ([3, H_b, W_b]) at the start of
 *   its slot of 3 * Hmax * Wmax bytes of `out` (rest of the slot zero).  Semantics per sample = memhip_rasterize_aug_f64;
 *   workspace = memhip_rasterize_workspace(B, Hmax, Wmax). */
int memhip_aug_resolve(const double* extent, memhip_event_aug_t* aug, int B, int stage, int fixed_h, int fixed_w,
                       int hmax, int wmax, int32_t* dims, int32_t* status, memhip_stream_t stream);
int memhip_rasterize_var_f64(const double* ev, const int64_t* offsets, const memhip_event_aug_t* aug,
                             const int32_t* dims, int B, int Hmax, int Wmax, int time_surface, uint8_t* out,
                             int32_t* status, void* workspace, size_t workspace_bytes, memhip_stream_t stream);

/* Per-sample extent of the (augmented) events, for the reference's data-dependent
 * canvas "W = xs.max()+1" (mem/datasets.py:516,538-540,572-575).
 * extent   f64 [B,4] out: max x, max y, min x, min y after `aug` (NULL = raw)
 *          and the bounds filter; rows of empty samples are (-inf,-inf,+inf,+inf). */
int memhip_events_extent(const double* ev, const int64_t* offsets, const memhip_event_aug_t* aug,
                         int B, double* extent, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Event record contract: raw dataset records -> the (N,4) float64 rows [x, y, t, p]
 * replaces the N-Caltech101 decode loop        process_data/process_dataset.py:48-63
 *          imgnet_npy_loader                   mem/dataset_folder.py:285-292
 *          dsec_npy_loader                     mem/dataset_folder.py:275-283
 * ------------------------------------------------------------------------
 * decode_ncaltech101: raw u8 [n_bytes] (device, 4-byte aligned) of 5-byte records: byte0 -> column 0,
 *   byte1 -> column 1, p = bit 7 of byte2 -> 2p-1, t = (byte2 & 0x7f):byte3:byte4 big-endian (23 bits);
 *   ev f64 [n_bytes/5, 4].  n_bytes % 5 != 0 -> MEMHIP_EINVAL (the reference's loop raises on the short read).
 * events_from_columns: four device column arrays of n elements with a dtype code each ->
 *   ev[i] = [x[i], y[i], t[i], int8(int8(p[i]) * 2 - 1)] as float64 (the int8 wrap-around of
 *   `data['p'].astype(np.int8) * 2 - 1` kept; p must be bool / integer).
 * events_dsec: in [n,4] row-major of `dtype` -> out f64 rows [x, y, t, 2p-1] of the rows with y < y_limit
 *   (440 in the reference), order kept; n_out i64 (device) = number of rows written; out has room for n rows. */
#define MEMHIP_DT_U8 0
#define MEMHIP_DT_I8 1
#define MEMHIP_DT_U16 2
#define MEMHIP_DT_I16 3
#define MEMHIP_DT_U32 4
#define MEMHIP_DT_I32 5
#define MEMHIP_DT_U64 6
#define MEMHIP_DT_I64 7
#define MEMHIP_DT_F32 8
#define MEMHIP_DT_F64 9
#define MEMHIP_DT_BOOL 10
int memhip_decode_ncaltech101(const uint8_t* raw, int64_t n_bytes, double* ev, memhip_stream_t stream);
int memhip_events_from_columns(const void* x, int x_dtype, const void* y, int y_dtype, const void* t, int t_dtype,
                               const void* p, int p_dtype, int64_t n, double* ev, memhip_stream_t stream);
size_t memhip_events_dsec_workspace(int64_t n);
int memhip_events_dsec(const void* in, int dtype, int64_t n, double y_limit, double* out, int64_t* n_out,
                       void* workspace, size_t workspace_bytes, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Image-space augmentation chain, batched with per-sample parameters (host draws, device arithmetic)
 * replaces ToTensor + Resize(bilinear, antialias=True) / RandomCrop(pad_if_needed)   mem/datasets.py:637-642
 *          ToUnit8 -> EventRandAugment(num_ops=2, 14 ops) -> ToFloat32               mem/datasets.py:655-658,
 *                                                                                    mem/transforms.py:292-484
 *          ColorJitter(brightness, 0, saturation)                                    mem/datasets.py:34-38
 * (torchvision tensor ops in the reference: uint8 images, float32 arithmetic, truncating casts, torch.round after
 * resampling; restated in oracle/aug_t.py.)
 * ------------------------------------------------------------------------
 * resample_to_f32: in u8 = B slots of slot_bytes, sample b stored densely as [3, h_b, w_b] (dims i32 [B,2], or
 *   fixed_h / fixed_w when dims is NULL) -> out f32 [B,3,OH,OW] = value / 255 resampled:
 *   mode 0 Resize((OH,OW), BILINEAR, antialias=True) per sample (separable triangle filter, horizontal first);
 *   mode 1 crop window (OH,OW) at offs[b] = (top, left) (NULL = 0,0) of the image zero-padded on BOTH sides by the
 *          deficit when it is smaller than the window (RandomCrop(pad_if_needed=True)).
 * to_uint8: y = (255 * x).to(uint8)  (ToUnit8, transforms.py:341-348).
 * rand_augment_u8: one op per sample on u8 [B,3,H,W] (out != in); ops = device array of B records
 *   { int32 op (index into ['Identity','ShearX','ShearY','TranslateX','TranslateY','Rotate','Brightness','Color',
 *   'Contrast','Sharpness','Posterize','Solarize','AutoContrast','Equalize']); float mag (Posterize bits / Solarize
 *   threshold); float theta[6] }: affine ops: theta = torchvision's inverse affine matrix as float32; blend ops
 *   (Brightness/Color/Contrast/Sharpness): theta[1] = float32(ratio), theta[0] = float32(1 - ratio), ratio = 1 + magnitude.
 * color_jitter: in u8 (value / 255 first = ToFloat32) or f32 [B,3,H,W] -> out f32 [B,out_chans,H,W] (2 = [pos,neg]);
 *   params = device array of B records { int32 order (0 none, 1 brightness, 2 saturation, 3 b then s, 4 s then b);
 *   float bf, 1-bf, sf, 1-sf } or NULL (conversion / channel drop only). */
int memhip_resample_to_f32(const uint8_t* in, const int32_t* dims, int64_t slot_bytes, int fixed_h, int fixed_w, int mode,
                           const int32_t* offs, int B, int OH, int OW, float* out, memhip_stream_t stream);
int memhip_to_uint8(const float* x, int64_t n, uint8_t* y, memhip_stream_t stream);
int memhip_rand_augment_u8(const uint8_t* in, uint8_t* out, const void* ops, int B, int H, int W, memhip_stream_t stream);
int memhip_color_jitter(const void* in, int in_is_u8, int B, int H, int W, const void* params, float* out, int out_chans,
                        memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Tensor-level event transforms, fused
 * replaces ToTensor (/255), RemoveTimesurface, RemoveHotPixels, LogTransform,
 * GammaTransform, NormalizeEvent            mem/transforms.py:200-275
 * in the order of build_transformNPY        mem/datasets.py:637-653
 * ------------------------------------------------------------------------
 * in        u8 [B,3,H,W] (in_is_u8=1: value/255 first) or f32 [B,3,H,W]
 * out       f32 [B, out_chans, H, W]; out_chans = 3 keeps [pos,tss,neg],
 *           out_chans = 2 keeps [pos,neg] (= x[:, 0::2], the "2-bin voxel")
 * flags     bit0 remove_timesurface, bit1 remove_hot_pixels, bit2 log,
 *           bit3 gamma, bit4 normalize
 */
#define MEMHIP_EV_RM_TS 1
#define MEMHIP_EV_HOTPIX 2
#define MEMHIP_EV_LOG 4
#define MEMHIP_EV_GAMMA 8
#define MEMHIP_EV_NORMALIZE 16
int memhip_event_norm(const void* in, int in_is_u8, int B, int H, int W, int flags,
                      float num_stds, float gamma, float* out, int out_chans,
                      memhip_stream_t stream);
/* RemoveHotPixels(num_hot_pixels = k) form of the same chain     mem/transforms.py:257-263
 * (the k largest entries of x[0::2].flatten() are hot; k is clamped to sum / 4 like the
 * reference; both polarities of a hot pixel are zeroed).  Equal values at the selection
 * boundary: the reference leaves their order to torch.argsort(stable=False); here the
 * larger flat index is taken.  hot_keys: u64 [B] scratch (the per-sample selection key).
 * MEMHIP_EV_HOTPIX is implied. */
int memhip_event_norm_topk(const void* in, int in_is_u8, int B, int H, int W, int flags,
                           int num_hot_pixels, float gamma, float* out, int out_chans,
                           uint64_t* hot_keys, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Mask generators (HOST code: bit-exact CPython `random` semantics need
 * glibc exp/log/sqrt and a sequential MT19937 stream)
 * replaces MaskingGenerator.__call__         mem/masking_generator.py:44-81
 *          MaskingGeneratorRandomLocation    mem/masking_generator.py:106-116
 * ------------------------------------------------------------------------
 * mt_state  host u32 [625]: the 624 MT19937 words + position, exactly the
 *           tuple random.getstate()[1]; updated in place.
 * out       host u8 [n_masks, H, W] (0/1)
 */
int memhip_mt_seed(uint32_t* mt_state, const uint32_t* key, int key_len);
double memhip_mt_random(uint32_t* mt_state);
int memhip_mask_blockwise(uint32_t* mt_state, int H, int W, int num_masking_patches,
                          int min_num_patches, int max_num_patches, double log_aspect_lo,
                          double log_aspect_hi, int n_masks, uint8_t* out);
int memhip_mask_random_location(uint32_t* mt_state, int H, int W, int num_masking_patches,
                                int n_masks, uint8_t* out);

/* ------------------------------------------------------------------------
 * bf16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  (+ fused epilogue)
 * replaces every F.linear / nn.Linear / nn.Conv2d(k=s=patch) of the ViT:
 *   patch embed  mem/modeling_finetune.py:203,209   qkv  :132-133   proj :155
 *   fc1+GELU     :61-68    fc2 :69    lm_head  mem/modeling_pretrain.py:59,126
 * and their dgrad / wgrad products in backward.
 * ------------------------------------------------------------------------
 * A, B are bf16 row-major with the reduction dimension contiguous (lda, ldb in
 * elements, multiples of 8; K a multiple of 64).  fp32 accumulate.  Output
 * rows >= M / cols >= N are never written.
 */
#define MEMHIP_EPI_BIAS_BF16 0   /* out0 bf16 = bf16(acc+bias); cols < colscale_n then *= colscale (q*scale, :137) */
#define MEMHIP_EPI_BIAS_GELU 1   /* out0 bf16 = h = bf16(acc+bias); out1 bf16 = gelu(h)       (:66-68) */
#define MEMHIP_EPI_RESIDUAL 2    /* y=bf16(acc+bias) -> out0 (may be NULL); resid f32 = (aux f32 or resid) + drop_path(vec1*y) (:187-188) */
#define MEMHIP_EPI_DGELU 3       /* out0 bf16 = bf16(acc) * gelu'(aux bf16)                   (GELU backward) */
#define MEMHIP_EPI_F32 4         /* out0 f32 (+)= acc                                         (weight gradients) */
#define MEMHIP_EPI_PATCH_EMBED 5 /* resid f32[b*(L+1)+1+p] = bf16(acc+bias)*(1-w) + vec1*w    (modeling_pretrain.py:101-108) */
#define MEMHIP_EPI_BIAS_GELU_DG 6 /* as BIAS_GELU, but out0 = gelu'(h) as FP16 (16 bits per value like bf16, 11 significant bits:
                                    gelu' lies in [-0.13, 1.13]) instead of h: the GELU backward then is a plain product
                                    (MUL_AUX) -- erf/exp are evaluated once, in the forward epilogue */
#define MEMHIP_EPI_MUL_AUX 7     /* out0 bf16 = bf16(acc) * aux fp16  (+ colsum)               (GELU backward with stored gelu') */
typedef struct memhip_gemm_args {
  const void* A; const void* B;
  int64_t lda, ldb;
  int32_t M, N, K, epilogue;
  void* out0; int64_t ldo0;
  void* out1; int64_t ldo1;
  const float* bias;      /* [N] fp32 or NULL */
  const float* vec1;      /* RESIDUAL: layer-scale gamma [N] (NULL = none); PATCH_EMBED: mask_token [N] */
  float* resid; int64_t ldr;   /* fp32 residual stream */
  const void* aux; int64_t ldaux; /* DGELU: pre-activation bf16 [M,N]; PATCH_EMBED: mask u8 [M];
                                     RESIDUAL: residual INPUT f32 [M,N] (NULL = resid in place) */
  const float* rowmask;   /* RESIDUAL: stochastic-depth keep mask per sample (0/1), NULL = off */
  float keep_prob;        /* RESIDUAL: 1 - drop_prob */
  float colscale; int32_t colscale_n;
  int32_t rows_per_sample;  /* RESIDUAL: tokens per sample; PATCH_EMBED: patches per sample */
  int32_t accumulate;     /* F32: 1 = out0 += acc */
  float* colsum;          /* BIAS_BF16 / DGELU: f32 [N] += column sums of the bf16 output rows < M (the
                             bias gradient of the Linear whose grad_output this GEMM produces); NULL = off */
  const int32_t* sample_map; /* RESIDUAL, stochastic depth as WORK SKIPPING (the GEMM runs on the rows of the kept samples
                             only): output row m belongs to compact sample c = m / rows_per_sample, and its residual row in
                             `resid` / `aux` is sample_map[c] * rows_per_sample + m % rows_per_sample.  The kept branch is
                             scaled by 1 / keep_prob; rowmask must be NULL.  The array must be readable for 256 entries
                             past the last compact sample.  NULL = rows map to themselves. */
  int32_t colsum_copies;  /* > 1: `colsum` holds that many accumulator copies of N floats each (copy k at colsum + k * N;
                             all zero on entry) and a workgroup adds into copy blockIdx % copies: atomics on one address
                             serialise (~0.17 us each on gfx950), and with one copy the ~400 per column of a ViT-B
                             launch sit in the in-order vector-memory queue in front of the operand stream (measured:
                             +27 us on the 3072-wide GELU' GEMM, +100 us on a 768 x 768 one).  Fold the copies with
                             memhip_colsum_fold.  0 / 1: a single accumulator (the bias gradient itself). */
  int32_t reserved0;
} memhip_gemm_args_t;
int memhip_gemm_bf16_nt(const memhip_gemm_args_t* args, memhip_stream_t stream);

/* Weight-gradient GEMM  out[N,K] (+)= sum_r A[r,N] * B[r,K]  (A = dY, B = X, both token-major
 * bf16 with the reduction over ROWS): the dY^T @ X that autograd computes for every Linear /
 * Conv2d weight on the path.  No transposed copies: fragments are read with the transposing LDS
 * read; split-K over the rows with fp32 atomics.  accumulate=1: add into `out` (pre-zeroed by
 * the caller); accumulate=0: overwrite. */
int memhip_gemm_bf16_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int R, int N, int K,
                        float* out, int64_t ldo, int accumulate, memhip_stream_t stream);
/* Same product with a caller-owned scratch buffer of memhip_gemm_bf16_tn_workspace(R,N,K) bytes (0 = the
 * shape has no use for one): the per-slice partial tiles are then written with plain stores and summed
 * by a reduction pass instead of fp32 atomics (64 MB of atomics per ViT-B weight gradient otherwise).
 * workspace NULL / too small: identical to memhip_gemm_bf16_tn.  The sum order is fixed, so the result
 * is run-to-run deterministic. */
size_t memhip_gemm_bf16_tn_workspace(int R, int N, int K);
int memhip_gemm_bf16_tn_ws(const void* A, int64_t lda, const void* B, int64_t ldb, int R, int N, int K,
                           float* out, int64_t ldo, int accumulate, void* workspace, size_t workspace_bytes,
                           memhip_stream_t stream);
/* The weight gradients of up to 4 Linear layers whose operands are ready at the same time (fc2 + fc1, proj + qkv of a
 * Block: mem/modeling_finetune.py:160-189 backward) as ONE launch: the products share one grid and one split count (the fewest rounds of workgroups that keep >= 80 % of the CUs busy), so a
 * small product (768 x 768: 9 tiles) runs with the 7 row slices of its neighbour instead of the 28 it needs alone to fill
 * the chip, and the group has one reduction pass.  Each product has the contract of memhip_gemm_bf16_tn_ws (fixed sum order:
 * run-to-run deterministic; the order differs from the single call's, so the two agree to fp32 rounding, not bitwise).
 * workspace: memhip_gemm_bf16_tn_group_workspace(problems, count) bytes (also >= what each product needs alone).  When the
 * group cannot run as one grid (count == 1, a shape outside the 256 x 256 tile kernel, workspace too small, option
 * "tn_group" = 0) the products are computed one after the other by memhip_gemm_bf16_tn_ws -- same results contract. */
typedef struct memhip_tn_problem {
  const void* A; int64_t lda;       /* dY  bf16 [R, N] */
  const void* B; int64_t ldb;       /* X   bf16 [R, K] */
  float* out; int64_t ldo;          /* dW  f32  [N, K] */
  int32_t R, N, K;
  int32_t reserved0;
} memhip_tn_problem_t;
size_t memhip_gemm_bf16_tn_group_workspace(const memhip_tn_problem_t* problems, int count);
int memhip_gemm_bf16_tn_group(const memhip_tn_problem_t* problems, int count, int accumulate, void* workspace,
                              size_t workspace_bytes, memhip_stream_t stream);
/* out f32 [C] += column sums of in bf16 [R, C]  (Linear bias gradients = grad_output.sum(0)) */
int memhip_colsum_bf16(const void* in, int64_t ld, int R, int C, float* out, memhip_stream_t stream);
/* out[n] += sum_k ws[k * N + n] for the `copies` accumulator copies a GEMM with colsum_copies > 1 filled; the copies
 * are zeroed again (ready for the next GEMM).  Same reduction order every run. */
int memhip_colsum_fold(float* ws, int copies, int N, float* out, memhip_stream_t stream);

/* Dead-row elimination in the LAST block (Block.forward, mem/modeling_finetune.py:160-189; the head reads only the masked
 * tokens' rows, mem/modeling_pretrain.py:119-126): the MLP branch of the last block runs on those rows in compact form.
 * residual_rows: out[i, :] = x[rows[i], :] + drop_path(gamma * y[i, :]) with y the bf16 output of the fc2 GEMM (+ bias) for
 * compact row i -- the arithmetic of the RESIDUAL epilogue; gamma NULL = no layer scale; rowkeep f32 [R] (0/1 per compact
 * row, NULL = no stochastic depth) with keep_prob.  scatter_rows: dst[rows[i], :] = src[i, :] (fp32 rows of D values). */
int memhip_residual_rows(const float* x, int64_t ldx, const int32_t* rows, const void* y, int64_t ldy, const float* gamma,
                         const float* rowkeep, float keep_prob, int R, int D, float* out, int64_t ldo,
                         memhip_stream_t stream);
int memhip_scatter_rows_f32(const float* src, int64_t lds, const int32_t* rows, int R, int D, float* dst, int64_t ldd,
                            memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * LayerNorm (eps 1e-6) forward / backward          mem/modeling_pretrain.py:132,
 * mem/modeling_finetune.py:166,172,184-188; final norm mem/modeling_pretrain.py:117
 * ------------------------------------------------------------------------
 * x fp32 [*, D] (residual stream), y bf16 [R, D] (what the next Linear consumes
 * under autocast), mean/rstd fp32 [R].  row_idx (i32 [R], may be NULL) selects
 * the input row of output row r: the final norm + `x[:,1:][bool_masked_pos]`
 * (modeling_pretrain.py:121-126) is one gathered LayerNorm over the masked rows.
 * backward: dres[row] (+)= dx, dgamma/dbeta fp32 [D] are ACCUMULATED.
 */
int memhip_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_idx, int R, int D,
                         const float* gamma, const float* beta, float eps, void* y_bf16, int64_t ldy,
                         float* mean, float* rstd, memhip_stream_t stream);
int memhip_layernorm_bwd(const void* dy_bf16, int64_t lddy, const float* x, int64_t ldx,
                         const int32_t* row_idx, int R, int D, const float* gamma, const float* mean,
                         const float* rstd, float* dres, int64_t lddres, int accumulate, float* dgamma,
                         float* dbeta, memhip_stream_t stream);

/* Layer-scale gradient from the weight gradient of the Linear that produced the branch:
 *   dgamma[c] = (sum_k W[c,k] * dW[c,k] + bias[c] * dbias[c]) / gamma[c]      (0 where gamma[c] == 0)
 * (x += gamma * y with y = A W^T + b: sum_m dt*y = sum_m dY*y / gamma and dW = dY^T A, dbias = colsum dY), so the
 * forward does not store y and memhip_branch_bwd / memhip_layernorm_bwd_branch run with y = NULL, dgamma = NULL.
 * W bf16 [N, ldw] (the operand the forward used), dW / dbias f32 = the gradients accumulated so far;
 * dgamma is OVERWRITTEN (it is a function of the accumulated dW). */
int memhip_layerscale_grad(const void* W_bf16, int64_t ldw, const float* dW, int64_t lddw, const float* bias,
                           const float* dbias, const float* gamma, int N, int K, float* dgamma,
                           memhip_stream_t stream);

/* memhip_layernorm_bwd (accumulating, no row gather) fused with the memhip_branch_bwd that follows it in a
 * block's backward: the updated dres row is consumed in registers (one pass over the fp32 gradient stream
 * less).  Arguments = those of the two calls; D <= 1024. */
int memhip_layernorm_bwd_branch(const void* dy_bf16, int64_t lddy, const float* x, int64_t ldx, int R, int D,
                                const float* gamma, const float* mean, const float* rstd, float* dres,
                                int64_t lddres, float* dgamma, float* dbeta, const void* y_branch_bf16, int64_t ldyb,
                                const float* gamma_branch, const float* rowmask, float keep_prob,
                                int rows_per_sample, void* dy_branch_bf16, int64_t lddyb, float* dgamma_branch,
                                float* dbias_branch, memhip_stream_t stream);

/* Backward of `x = x + drop_path(gamma * y)` (mem/modeling_finetune.py:187-188):
 * dy bf16 = bf16(dt * gamma), dgamma += sum_m dt*y, dbias += sum_m dy with
 * dt = dx * rowmask[m / rows_per_sample] / keep_prob.  gamma/rowmask/dgamma/dbias may be NULL; y may be NULL
 * when dgamma is (see memhip_layerscale_grad). */
int memhip_branch_bwd(const float* dx, int64_t lddx, const void* y_bf16, int64_t ldy, const float* gamma,
                      const float* rowmask, float keep_prob, int rows_per_sample, int M, int D,
                      void* dy_bf16, int64_t lddy, float* dgamma, float* dbias, memhip_stream_t stream);
/* Stochastic depth as WORK SKIPPING (timm drop_path, mem/modeling_finetune.py:42-53,187-188: a dropped sample's branch
 * contributes nothing, so nothing of it is computed).  Sample maps, i32 [samples]: sample -> its index among the samples the
 * branch KEPT this step, or -1.  memhip_branch_bwd_map: out_map places the kept samples' rows of dy compactly (row
 * out_map[s] * rows_per_sample + t) scaled by 1 / keep_prob; dropped samples are neither read nor written.
 * memhip_layernorm_bwd_branch_map: in_map says which samples the LayerNorm'ed branch kept (dy / mean / rstd hold those
 * samples only; the others get no LayerNorm gradient), out_map the same for the branch whose output gradient is produced.
 * M / R count the rows of the residual stream (all samples).  rowmask and y must be NULL; NULL maps = identity. */
int memhip_branch_bwd_map(const float* dx, int64_t lddx, const void* y_bf16, int64_t ldy, const float* gamma,
                          const float* rowmask, float keep_prob, int rows_per_sample, int M, int D,
                          void* dy_bf16, int64_t lddy, float* dgamma, float* dbias, const int32_t* out_map,
                          memhip_stream_t stream);
int memhip_layernorm_bwd_branch_map(const void* dy_bf16, int64_t lddy, const float* x, int64_t ldx, int R, int D,
                                    const float* gamma, const float* mean, const float* rstd, float* dres,
                                    int64_t lddres, float* dgamma, float* dbeta, const void* y_branch_bf16, int64_t ldyb,
                                    const float* gamma_branch, const float* rowmask, float keep_prob,
                                    int rows_per_sample, void* dy_branch_bf16, int64_t lddyb, float* dgamma_branch,
                                    float* dbias_branch, const int32_t* in_map, const int32_t* out_map,
                                    memhip_stream_t stream);

/* Backward of the token assembly (mem/modeling_pretrain.py:101-108): dcls += dx[cls rows],
 * dmask_token += sum dx*w, dy bf16 [B*L, D] = bf16(dx*(1-w)); mask u8 [B*L]. */
int memhip_embed_bwd(const float* dx, int64_t lddx, const uint8_t* mask, int B, int L, int D,
                     void* dy_bf16, int64_t lddy, float* dcls, float* dmask_token, memhip_stream_t stream);

/* nn.CrossEntropyLoss()(logits, labels) + mlm_acc  (mem/engine_for_pretraining.py:152,233).
 * logits bf16 [M, V] (in place -> dlogits = (softmax - onehot) * grad_scale when write_grad);
 * row_loss f32 [M], row_correct i32 [M] scratch; out2 f32 [2] = {mean loss, accuracy}. */
int memhip_cross_entropy(void* logits_bf16, int64_t ld, const int64_t* labels, int M, int V,
                         float grad_scale, float* row_loss, int32_t* row_correct, int write_grad,
                         float* out2, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused attention with shared relative position bias, head_dim 64, <= 256 tokens
 * replaces Attention.forward q*scale .. (attn@v)   mem/modeling_finetune.py:137-154
 *          RelativePositionBias (index + forward)    mem/modeling_finetune.py:213-247
 * ------------------------------------------------------------------------
 * qkv    bf16 [B*T, 3D] token-major, columns [q*scale | k | v], head h at h*64
 * table  f32 [(2Wh-1)(2Ww-1)+3, heads] = relative_position_bias_table; T = Wh*Ww + 1 (cls first).
 *        The additive bias is gathered from the table on chip (bucket index computed
 *        arithmetically, identical to relative_position_index); no [heads,T,T] tensor is read.
 * out    bf16 [B*T, D];  lse f32 [B, heads, TP], TP = memhip_attn_tokens_padded(T) (for backward)
 * bwd:   delta f32 [(2*B*T + 4) * heads]: memhip_attn_delta fills [B*T, heads] = rowsum(dout * out)
 *        per head and, behind it, [B*T, heads] = |dout_row,head|^2; the last 4 * heads floats are
 *        scratch of memhip_attn_bwd (per-head bounds max |dout_row|^2, max |delta|, max |v_key|^2
 *        that scale the fixed-point buckets of the table gradient);
 *        dqkv bf16 [B*T, 3D] (dq already multiplied by `scale`);
 *        dtable f32 [num_rel, heads] += table gradient (NULL skips it);
 *        dq_bias / dv_bias f32 [D] += column sums of dq / dv (the q_bias / v_bias gradients).
 * memhip_relpos_gather materialises the bias tensor (tests / inspection only).
 */
int memhip_attn_tokens_padded(int T);
int memhip_relpos_gather(const float* table, const int32_t* index /*[T*T]*/, int T, int TP, int heads,
                         float* bias_pad, float* biasT_pad /*[heads,TP(key),TP(query)] or NULL*/,
                         memhip_stream_t stream);
int memhip_attn_fwd(const void* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table,
                    int window_h, int window_w, void* out, int64_t ldo, float* lse, memhip_stream_t stream);
int memhip_attn_delta(const void* dout, const void* out, int64_t ldo, int64_t rows, int heads, float* delta,
                      memhip_stream_t stream);
int memhip_attn_bwd(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse,
                    float* delta, const float* table, int window_h, int window_w, int B, int T, int D,
                    int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias,
                    float* dv_bias, memhip_stream_t stream);
/* The same backward given the forward OUTPUT `out` (bf16 [B*T, D], leading dimension ldout) instead of a filled `delta`:
 * rowsum(dout * out) is computed by the library -- inside the fused 14 x 14 kernel when that kernel applies (no separate
 * pass over dout and out: 25 us per ViT-B layer at B = 256), otherwise by memhip_attn_delta into `delta` (then
 * ldout == ldo is required).  `delta` is still the [(2*B*T + 4) * heads] workspace.  (Attention.forward backward,
 * mem/modeling_finetune.py:137-154.) */
int memhip_attn_bwd_out(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout,
                        const float* lse, float* delta, const float* table, int window_h, int window_w, int B, int T,
                        int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias,
                        float* dv_bias, memhip_stream_t stream);
/* The same two calls with a caller-owned WORKSPACE (round 6).  For the long windows the slot-layout kernels take (40 or 20 tokens
 * wide, more than 256 tokens: BASELINE configs[4]) the backward then runs in its dS-storing form: the dK / dV kernel writes
 * dS (bf16) to the workspace and owns the table gradient, the dQ kernel is a streaming product over it -- the score tile is
 * computed once instead of twice.  memhip_attn_bwd_workspace returns the bytes that form wants (0: this shape has no such
 * form); ws == NULL, too few bytes or any other shape = exactly memhip_attn_bwd / memhip_attn_bwd_out.  The workspace is
 * scratch: nothing is kept in it between calls.  Same outputs and rounding points either way (the table gradient is summed
 * in another order: fixed-point buckets per workgroup, then float atomics).  (Attention.forward backward,
 * mem/modeling_finetune.py:137-154, RelativePositionBias :213-247.) */
int64_t memhip_attn_bwd_workspace(int B, int T, int heads, int window_h, int window_w);
int memhip_attn_bwd_ws(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const float* lse,
                       float* delta, const float* table, int window_h, int window_w, int B, int T, int D,
                       int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias,
                       float* dv_bias, void* ws, int64_t ws_bytes, memhip_stream_t stream);
int memhip_attn_bwd_out_ws(const void* qkv, int64_t ldqkv, const void* dout, int64_t ldo, const void* out, int64_t ldout,
                           const float* lse, float* delta, const float* table, int window_h, int window_w, int B, int T,
                           int D, int heads, float scale, void* dqkv, int64_t lddqkv, float* dtable, float* dq_bias,
                           float* dv_bias, void* ws, int64_t ws_bytes, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * fp32 PARITY MODE (`--precision fp32`): the ViT path with fp32 operands / accumulation and no bf16 rounding points --
 * the reference's arithmetic without autocast (same reference lines as the bf16 entry points above).  For loss-curve
 * parity against the reference's fp32 CPU run (north star: step-100 loss within 1e-4); speed is secondary.
 * ------------------------------------------------------------------------
 * f32_gemm_nt: memhip_gemm_args_t with A, B, out0, out1 and (DGELU) aux as fp32; epilogues BIAS_BF16 (= plain bias,
 *   fp32 out), BIAS_GELU (exact erf), RESIDUAL, DGELU, F32, PATCH_EMBED; K and ld multiples of 4 (K is zero-extended to
 *   the next multiple of 32 inside).  f32_transpose: out [C, ldout] = in [R, C]^T, columns >= R zero.
 * f32_attn_*: generic attention (head_dim 32 or 64, T <= 256): bias[h,i,j] = table[index[i*T+j], h] (index NULL: none);
 *   bwd zeroes dqkv, then dq (times scale) by stores, dk / dv / dtable by fp32 atomics; delta = sum_j P dP. */
int memhip_f32_gemm_nt(const memhip_gemm_args_t* args, memhip_stream_t stream);
int memhip_f32_transpose(const float* in, int64_t ldin, int R, int C, float* out, int64_t ldout, memhip_stream_t stream);
int memhip_f32_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_idx, int R, int D, const float* gamma,
                             const float* beta, float eps, float* y, int64_t ldy, float* mean, float* rstd,
                             memhip_stream_t stream);
int memhip_f32_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const int32_t* row_idx, int R, int D,
                             const float* gamma, const float* mean, const float* rstd, float* dres, int64_t lddres,
                             int accumulate, float* dgamma, float* dbeta, memhip_stream_t stream);
int memhip_f32_branch_bwd(const float* dx, int64_t lddx, const float* y, int64_t ldy, const float* gamma, const float* rowmask,
                          float keep_prob, int rows_per_sample, int M, int D, float* dy, int64_t lddy, float* dgamma,
                          float* dbias, memhip_stream_t stream);
int memhip_f32_embed_bwd(const float* dx, int64_t lddx, const uint8_t* mask, int B, int L, int D, float* dy, int64_t lddy,
                         float* dcls, float* dmask_token, memhip_stream_t stream);
int memhip_f32_cross_entropy(float* logits, int64_t ld, const int64_t* labels, int M, int V, float grad_scale,
                             float* row_loss, int32_t* row_correct, int write_grad, float* out2, memhip_stream_t stream);
int memhip_f32_colsum(const float* in, int64_t ld, int R, int C, float* out, memhip_stream_t stream);
int memhip_f32_im2col(const float* x, int B, int C, int H, int W, int ph, int pw, float* out, memhip_stream_t stream);
int memhip_f32_attn_fwd(const float* qkv, int64_t ldqkv, int B, int T, int D, int heads, const float* table,
                        const int32_t* index, float* out, int64_t ldo, float* lse, memhip_stream_t stream);
int memhip_f32_attn_bwd(const float* qkv, int64_t ldqkv, const float* dout, int64_t ldo, int B, int T, int D, int heads,
                        float scale, const float* table, const int32_t* index, float* dqkv, int64_t lddqkv, float* dtable,
                        memhip_stream_t stream);

/* MAE variant (`--mae 1`), fp32 path: token plumbing and loss around the generic blocks above
 * replaces random_masking gather / mask-token unshuffle / forward_loss     mem/modeling_mae.py:204-292
 *   enc_assemble: out [B,(K+1),D]: row (b,0) = cls + pos[0]; row (b,1+j) = xe[b, ids_keep[b,j]] + pos[1 + ids_keep[b,j]]
 *   dec_assemble: out [B,(L+1),D]: row (b,0) = y[b,0] + dpos[0]; row (b,1+l) = (r < K ? y[b,1+r] : mask_token) + dpos[1+l],
 *                 r = ids_restore[b,l]
 *   *_bwd: the exact transposes (dxe is zeroed inside; dcls / dmask_token accumulate)
 *   mae_loss: pred [B,(L+1),p*p*C] (cls row ignored) vs patchify(img) ('nchpwq->nhwpqc'); row_loss [B,L] = per-patch mean
 *             squared error times its weight (only_masked: mask / sum(mask), else 1), dpred = gradient of the summed loss;
 *             scratch2[0] = sum(mask), scratch2[1] = loss. */
int memhip_mae_enc_assemble(const float* xe, const float* pos, const float* cls, const int64_t* ids_keep, int B, int L, int K,
                            int D, float* out, memhip_stream_t stream);
int memhip_mae_enc_assemble_bwd(const float* dx, const int64_t* ids_keep, int B, int L, int K, int D, float* dxe, float* dcls,
                                memhip_stream_t stream);
int memhip_mae_dec_assemble(const float* y, const float* mask_token, const float* dpos, const int64_t* ids_restore, int B, int L,
                            int K, int D, float* out, memhip_stream_t stream);
int memhip_mae_dec_assemble_bwd(const float* dxd, const int64_t* ids_restore, int B, int L, int K, int D, float* dy,
                                float* dmask_token, memhip_stream_t stream);
int memhip_mae_loss(const float* pred, const float* img, const float* mask, int B, int C, int H, int W, int patch,
                    int only_masked, float* row_loss, float* dpred, float* scratch2, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Frozen dVAE tokenizer forward
 * replaces DiscreteVAE.get_codebook_indices          eventvae/vae/vae_model.py:153-158
 *          encoder stack / ResBlock                   eventvae/vae/vae_model.py:29-42,86-101
 * called every pretraining step at                   mem/engine_for_pretraining.py:144
 * ------------------------------------------------------------------------
 * Activations are bf16 NHWC with a one-pixel zero border: [B, H+2, W+2, C] (the caller zeroes the
 * buffers once; the kernels only write interiors).  conv2d: out = [relu](conv(in, weight) + bias) [+ add];
 * weight bf16 [C_out, k*k*C_in] packed (ky, kx, c)-major; shapes 4x4/s2/p1, 3x3/s1/p1, 1x1/s1/p0;
 * C_in = 4 (first layer: 3 channels + 1 zero, 4x4 only) or a multiple of 64; out_padded = 0 writes a
 * dense [B*Ho*Wo, C_out] matrix (the token logits).  `add` has the layout of `out` (ResBlock residual).
 * Precision: bf16 operands, fp32 accumulate (the reference runs this stage in fp32 / TF32). */
int memhip_conv2d_nhwc_bf16(const void* in, const void* weight, const float* bias, const void* add, void* out,
                            int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                            int out_padded, memhip_stream_t stream);
/* x f32 NCHW [B, C<=4, H, W] -> bf16 [B, H+2, W+2, 4] interior; mean/std f32 [C] or both NULL (DiscreteVAE.norm) */
int memhip_nchw_to_padded_nhwc4(const float* x, int B, int C, int H, int W, const float* mean, const float* stdv,
                                void* out, memhip_stream_t stream);
/* ids i64 [M] = argmax over the N columns of logits bf16 [M, ld] (first maximum) */
int memhip_argmax_rows_bf16(const void* logits, int64_t ld, int M, int N, int64_t* ids, memhip_stream_t stream);

/* The same three entry points in fp32 -- the EXACT-label mode (default): the reference computes the tokenizer in fp32
 * (mem/engine_for_pretraining.py:140-145 is outside the autocast block at :147) and its output is an integer, so the
 * labels are produced with fp32 operands and fp32 accumulation (v_mfma_f32_16x16x4_f32, an fmaf chain over k).
 * Activations fp32 NHWC with the one-pixel zero border, weight fp32 [C_out, k*k*C_in] (ky, kx, c)-major; any kernel
 * size 1..4, stride >= 1, pad 0/1; C_in, C_out multiples of 4, k*k*C_in a multiple of 32.
 * argmax_rows_f32: ids = first maximum of each row; top2_gap (f32 [M], may be NULL) = best - runner-up logit. */
int memhip_conv2d_nhwc_f32(const float* in, const float* weight, const float* bias, const float* add, float* out,
                           int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                           int out_padded, memhip_stream_t stream);
int memhip_nchw_to_padded_nhwc4_f32(const float* x, int B, int C, int H, int W, const float* mean, const float* stdv,
                                    float* out, memhip_stream_t stream);
int memhip_argmax_rows_f32(const float* logits, int64_t ld, int M, int N, int64_t* ids, float* top2_gap,
                           memhip_stream_t stream);

/* Certified split-precision tokenizer (round 5; replaces the same reference lines: eventvae/vae/vae_model.py:153-158, called
 * at mem/engine_for_pretraining.py:139-145 in fp32).  The fp16x2 convolutions below produce the logits; a label is ACCEPTED
 * only where its top-2 gap exceeds kappa x the row's rms (kappa = twice a stated bound on the fp16x2 logit deviation relative
 * to the row rms: then the fp32 argmax is the same index); every sample that holds a token below the margin is recomputed
 * on the fp32 path above and its labels are replaced.  Everything is decided on the device (no host synchronisation):
 *   argmax_rows_f32_ex   argmax_rows_f32 + row_rms (f32 [M], may be NULL) = sqrt(mean_n logit^2); with n_samples (device
 *                        int32, may be NULL) only the first *n_samples x rows_per_sample rows exist
 *   tok_flag_samples     list (int32 [B]) = indices of the samples with a token whose gap is not > kappa x rms (ascending),
 *                        count[0] = their number; stats (int64 [2], may be NULL): += flagged samples, += 1 call
 *   tok_gather_images_f32  nchw_to_padded_nhwc4_f32 of the samples list[offset + j], j < n_round[0] = clamp(count - offset, 0, R)
 *                        into slots 0.. of `out` (n_round is written: the dynamic batch of the round)
 *   conv2d_nhwc_f32_dyn  conv2d_nhwc_f32 on a buffer with capacity B of which only the first *n_active samples are live
 *                        (the grid covers the capacity; tiles behind the live rows return at once)
 *   tok_scatter_ids      ids_out[list[offset + j] * tokens_per_sample + t] = ids_in[j * tokens_per_sample + t], j < *n_round */
int memhip_argmax_rows_f32_ex(const float* logits, int64_t ld, int M, int N, int64_t* ids, float* top2_gap, float* row_rms,
                              const int32_t* n_samples, int rows_per_sample, memhip_stream_t stream);
int memhip_tok_flag_samples(const float* top2_gap, const float* row_rms, int B, int tokens_per_sample, float kappa,
                            int32_t* list, int32_t* count, int64_t* stats, memhip_stream_t stream);
int memhip_tok_gather_images_f32(const float* x, int C, int H, int W, const float* mean, const float* stdv,
                                 const int32_t* list, const int32_t* count, int offset, int R, float* out, int32_t* n_round,
                                 memhip_stream_t stream);
int memhip_conv2d_nhwc_f32_dyn(const float* in, const float* weight, const float* bias, const float* add, float* out,
                               int B, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu,
                               int out_padded, const int32_t* n_active, memhip_stream_t stream);
int memhip_tok_scatter_ids(const int64_t* ids_in, const int32_t* list, const int32_t* n_round, int offset, int R,
                           int tokens_per_sample, int64_t* ids_out, memhip_stream_t stream);

/* "fp16 x 2" mode (opt-in, `--tokenizer_impl hip_fp16x2`): every fp32 value travels as two fp16 planes hi = fp16(v),
 * lo = fp16((v - hi) * 2048); a product is three fp16 MFMAs (hi*hi + (hi*lo + lo*hi) / 2048, exact products, fp32
 * accumulation).  Logits within 1.4e-5 of the fp32 module at a spread of 1.77 (between the exact fp32 mode and the bf16
 * mode), ~2x faster than fp32.  Tensors: base pointer of the hi plane + plane stride in elements to the lo plane; weights
 * [2][C_out, k*k*C_in]; shapes as memhip_conv2d_nhwc_bf16; out_f32 = 1 writes the dense fp32 logit matrix. */
int memhip_conv2d_nhwc_f16x2(const void* in, int64_t in_plane, const void* weight, int64_t w_plane, const float* bias,
                             const void* add, int64_t add_plane, void* out, int64_t out_plane, int B, int H, int W, int Cin,
                             int Cout, int ksize, int stride, int pad, int relu, int out_padded, int out_f32,
                             memhip_stream_t stream);
int memhip_nchw_to_padded_nhwc4_f16x2(const float* x, int B, int C, int H, int W, const float* mean, const float* stdv,
                                      void* out, int64_t out_plane, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Layout / dtype movers
 * ------------------------------------------------------------------------ */
int memhip_cast_f32_bf16(const float* in, void* out_bf16, int64_t n, memhip_stream_t stream);
/* Zero fills of the training step (the reference: optimizer.zero_grad(), mem/engine_for_pretraining.py:160; torch.zeros
 * temporaries of autograd): `bytes` at p, or n ranges {byte offset, byte count} (device array of 2n int64, every value a
 * multiple of 16, total_bytes = their sum: it sizes the grid) relative to base, in one launch -- e.g. every gradient that is
 * accumulated by atomics, while the weight matrices are written, not accumulated, by the weight-gradient GEMM. */
int memhip_zero(void* p, int64_t bytes, memhip_stream_t stream);
int memhip_zero_ranges(void* base, const int64_t* ranges, int n, int64_t total_bytes, memhip_stream_t stream);
/* dst[s] = src[s] for the n listed samples s = ids[k] (n_per_sample fp32 values each, a multiple of 4): the residual rows of
 * the samples a stochastic-depth branch dropped (mem/modeling_finetune.py:187-188 with a zero keep mask) */
int memhip_copy_samples_f32(const float* src, float* dst, const int32_t* ids, int n, int64_t n_per_sample,
                            memhip_stream_t stream);
/* out bf16 [Cc, ldout] = in f32 [R, Cc]^T (the [in,out]-major copy of a Linear weight used by dgrad) */
int memhip_transpose_cast_f32_bf16(const float* in, int64_t ldin, int R, int Cc, void* out_bf16,
                                   int64_t ldout, memhip_stream_t stream);
/* n weight transposes (fp32 [R,C] -> bf16 [C, ldout], rows beyond R zero-filled up to min(ldout, 64-padded R))
 * in one launch: desc = device array of n records {const float* in; int64 ldin; int64 R; int64 C; bf16* out;
 * int64 ldout} (48 bytes each), tile_prefix = device i32 [n+1] with the first 64x64 tile index of every matrix
 * (tile_prefix[n] = total_tiles).  Same result as n calls of memhip_transpose_cast_f32_bf16. */
int memhip_transpose_cast_batched(const void* desc, const int32_t* tile_prefix, int n, int total_tiles,
                                  memhip_stream_t stream);

/* out bf16 [Cc, ldout] = in bf16 [R, Cc]^T, columns [R, R_pad) zero-filled (R_pad % 64 == 0);
 * optional fused column sums of `in` (Linear bias gradients) over two column ranges. */
int memhip_transpose_bf16(const void* in, int64_t ldin, int R, int Cc, void* out, int64_t ldout, int R_pad,
                          float* colsum0, int c0_begin, int c0_end, float* colsum1, int c1_begin,
                          int c1_end, memhip_stream_t stream);
/* im2col of the k=s=patch conv (mem/modeling_finetune.py:203-209): x f32 [B,C,H,W] ->
 * bf16 [B*L, C*ph*pw], k = c*ph*pw + py*pw + px */
int memhip_im2col_bf16(const float* x, int B, int C, int H, int W, int ph, int pw, void* out_bf16,
                       memhip_stream_t stream);
/* x[b*T, :] = cls_token (mem/modeling_pretrain.py:101,108) */
int memhip_fill_cls(float* x, int64_t ldx, int B, int T, int D, const float* cls, memhip_stream_t stream);
/* y[n] += sum_k W[n,k] x[k]  (W bf16 [N, ldw], x f32 [K], y f32 [N]); optionally x_acc[k] += x[k] and
 * zero[k] = 0 (K floats each, may be NULL; `zero` must not alias x).  Used for the v_bias gradient:
 * sum over keys of dV = sum over queries of d(attn_out) (softmax rows sum to one) = (column sums of the
 * proj-output gradient) @ W_proj, so v_bias.grad is one 768x768 GEMV per block (modeling_finetune.py:128-139). */
int memhip_gemv_bf16_acc(const void* W, int64_t ldw, int N, int K, const float* x, float* y, float* x_acc,
                         float* zero, memhip_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimizer step on flat fp32 buffers (tensors padded to 1024 elements)
 * replaces clip_grad_norm_ / get_grad_norm_   mem/utils.py:360-366,380-392
 *          optim.AdamW(betas=(0.9,0.95))      mem/optim_factory.py:121,132-133
 * ------------------------------------------------------------------------ */
size_t memhip_grad_norm_workspace(void);
int memhip_grad_norm(const float* g, int64_t n, float* norm_out, void* workspace, size_t workspace_bytes,
                     memhip_stream_t stream);
/* wd_flag_per_chunk u8 [n/1024]: 1 = weight decay applies to that chunk.  gnorm (device scalar) and
 * max_norm > 0 fold gradient clipping into the update; step >= 1 is the AdamW step count. */
int memhip_adamw(float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* wd_flag_per_chunk,
                 double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                 const float* gnorm, double max_norm, memhip_stream_t stream);

/* The same step with per-group learning rates / weight decay (layer-wise lr decay of finetuning:
 * replaces get_parameter_groups + LayerDecayValueAssigner   mem/optim_factory.py:31-100 feeding optim.AdamW).
 * group_of_chunk u8 [n/1024]; group_table f32 [n_groups][2] (device) = {1 - lr_g*wd_g, lr_g / (1 - beta1^step)},
 * computed by the caller in double and rounded once, like torch's Python-side scalars. */
int memhip_adamw_groups(float* p, const float* g, float* m, float* v, int64_t n, const uint8_t* group_of_chunk,
                        const float* group_table, int n_groups, double beta1, double beta2, double eps, int step,
                        const float* gnorm, double max_norm, memhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MEMHIP_H */
