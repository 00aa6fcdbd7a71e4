/* keep_hip.h -- C-ABI of libkeep_hip.so: the gfx950 (MI355X / CDNA4) kernels behind the KEEP
 * inference hot path.
 *
 * Boundary: the reference hot path is 100 % stock PyTorch ops -- it has NO native/FFI
 * layer of its own for this path -- so this ABI is defined by the build.  Each entry point names the
 * reference arithmetic it replaces (file:line under /root/reference/modules/deps/wm_basicsr/archs/;
 * KA = keep_arch.py, VQ = vqgan_arch.py, AU = arch_util.py, GM = gmflow/gmflow/).  The reference-side
 * binding a maintainer would add is the ctypes stub shown in INTEGRATION.md (engine/hiplib.py here).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensor.data_ptr()); the library never
 *     allocates, frees or retains memory; workspaces are caller-provided;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream), has no
 *     hidden synchronisation and no global mutable state -> capturable in a hipGraph, thread-safe per stream;
 *   - activations are channels-last: [N, H, W, C] (tokens [B, L, C] are the same memory);
 *   - returns 0 on success, KEEP_EINVAL (-1) bad argument/shape, KEEP_EUNSUP (-2) unsupported dtype/arch,
 *     KEEP_EHIP (-3) HIP runtime error; keep_last_error() gives a thread-local message.
 */
#ifndef KEEP_HIP_H
#define KEEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY layout or signature change.  v12: keep_conv2d_args / keep_attention_args start with `struct_size`
 * (callers set it to sizeof of the struct THEY were compiled against; the library rejects sizes it does not know and reads
 * the fields a shorter known layout lacks as zero), keep_sizeof_*_args(), keep_argmax_gather takes the non-finite status word,
 * keep_nonfinite_flag.  v13: keep_conv2d_args.upsample accepts KEEP_UPSAMPLE_X2_PHASES (same layout; a v12 library refuses the
 * value, so the binding asks for 13).  v18: the two reserved words of keep_conv2d_args become `flags` / `plan_ref_images`, the one of
 * keep_attention_args `flags` (same layout and sizes; zero keeps the v17 behaviour) -- the library no longer reads ANY environment variable.  v19: keep_yolo_letterbox_u8, keep_yolo_select, keep_layernorm_amax, keep_geglu_amax, keep_retina_nms_ordered (additions only). */
#define KEEP_ABI_VERSION 20
#define KEEP_OK 0
#define KEEP_EINVAL (-1)
#define KEEP_EUNSUP (-2)
#define KEEP_EHIP (-3)

/* storage dtypes of activation tensors */
#define KEEP_F32 0
#define KEEP_BF16 1

/* matrix-core operand precision (accumulation is always fp32):
 *   KEEP_MMA_F32  v_mfma_f32_32x32x2_f32   exact f32 products, 157 TFLOP/s peak  -- the <=1e-3 parity policy
 *   KEEP_MMA_BF16 v_mfma_f32_32x32x16_bf16 operands rounded to bf16 (RNE) when staged into LDS, 2.5 PFLOP/s peak;
 *                 activations stay fp32 in HBM, weights come from the caller's bf16 copy (`weight_bf16`) */
#define KEEP_MMA_F32 0
#define KEEP_MMA_BF16 1
/*   KEEP_MMA_X3   v_mfma_f32_32x32x16_f16 x 3: every fp32 operand split as hi + lo (two fp16), a*b = a_hi*b_hi + a_hi*b_lo +
 *                 a_lo*b_hi into one fp32 accumulator: <= 2^-22 relative per product (fp32-grade) at 2.5 PF / 3 = 833 TFLOP/s
 *                 -- the parity-grade FAST policy (csrc/keep_conv_x3.hip).  Activations are split on the fly; weights come
 *                 pre-scaled + pre-split from the caller (`weight_x3`, `x3_acc_scale`).  Geometries the x3 kernels do not
 *                 cover run on the exact-f32 kernels.  An activation beyond the fp16 range (65504) yields inf, never a
 *                 silently wrong value. */
#define KEEP_MMA_X3 2

/* prologue activation applied to the (affine-normalised) conv input */
#define KEEP_PRO_NONE 0
#define KEEP_PRO_SWISH 1 /* x*sigmoid(x)            VQ:20-22 */
#define KEEP_PRO_RELU 2  /* GM/backbone.py:30-31    */

/* epilogue activation applied to acc+bias */
#define KEEP_ACT_NONE 0
#define KEEP_ACT_RELU 1
#define KEEP_ACT_LRELU02 2 /* LeakyReLU(0.2)        KA:449,454 */
#define KEEP_ACT_GELU 3    /* exact erf GELU        KA:437, GM/transformer.py:141 */
#define KEEP_ACT_SIGMOID 4 /* KA:771 */
#define KEEP_ACT_SILU 6    /* x * sigmoid(x): Conv / ShuffleV2Block of the YOLOv5-face detectors, yolov5face/models/common.py:39-45,123-146 */
#define KEEP_ACT_LRELU01 5 /* LeakyReLU(0.1): MobileNetV1 / FPN / SSH of retinaface_mobile0.25, retinaface_net.py:6-34,41-43,74-76 */

/* keep_conv2d_args.flags (v18): kernel-selection overrides, 0 = the library's own choice.  The library reads no environment variable;
 * what used to be KEEP_NO_* / KEEP_X3_EXACT_ACT / KEEP_GATHER_SMALL_M switches travels with the call.  The NO_* bits take one kernel
 * family out of the dispatch (the next more general family runs): the kernel-vs-kernel parity tests and A/B runs use them. */
#define KEEP_CONV_NO_COUT4 (1u << 0)     /* Cout <= 4 exact-fp32 VALU kernel */
#define KEEP_CONV_NO_C3 (1u << 1)        /* Cin <= 3 first-convolution kernels */
#define KEEP_CONV_NO_HALO_F32 (1u << 2)  /* exact-f32 LDS-halo 3x3 kernel */
#define KEEP_CONV_NO_HALO_X3 (1u << 3)   /* x3 LDS-halo 3x3 kernels (both) */
#define KEEP_CONV_NO_GATHER_X3 (1u << 4) /* x3 implicit-GEMM kernel (-> exact-f32 kernels) */
#define KEEP_CONV_NO_PLAIN (1u << 5)     /* prologue-free fast staging of the gather kernels */
#define KEEP_CONV_NO_FLATK_F32 (1u << 6) /* flattened-K form for Cin < 8 */
#define KEEP_CONV_SMALL_TILES (1u << 7)  /* gather kernels: 64x64 tiles whatever the row count */
#define KEEP_CONV_X3_EXACT_ACT (1u << 8) /* x3 kernels: library expf / erff instead of the x3-grade fast forms */
#define KEEP_CONV_NO_STREAM (1u << 9)    /* x3 3x3: the stage-barrier-MFMA halo kernel instead of the streaming one */
#define KEEP_CONV_NO_GEMM_LAT (1u << 10) /* x3 GEMM form with <= 256 rows per image: one sequential sum (conv_x3_kernel) instead of canonical K slices */
#define KEEP_CONV_GEMM_LAT_WAVES (1u << 11) /* canonical K slices: the one-wave-per-slice kernel (gemm_x3l_kernel) whatever the row count (same bits) */
#define KEEP_CONV_NO_SMALL_PARTIALS (1u << 13) /* x3 3x3 split-K plans with few images: conv3x3_halo_x3_kernel's 256-pixel blocks instead of conv3x3_x3p_kernel (same partials) */
#define KEEP_CONV_GEMM_LAT_TILES (1u << 12) /* canonical K slices: conv_x3_kernel with slice totals whatever the row count (same bits) */
/* keep_attention_args.flags (v18) */
#define KEEP_ATTN_NO_PACK (1u << 0)   /* x3: never pre-pack K / V^T (keep_attention_workspace_bytes answers 0) */
#define KEEP_ATTN_NO_SFULL2 (1u << 1) /* x3, D = 512: the 128-query kernel instead of the 32-query one */
#define KEEP_ATTN_NO_X3 (1u << 2)     /* KEEP_MMA_X3 calls run on the exact-f32 kernel */
#define KEEP_ATTN_NO_TWO_PASS (1u << 4) /* x3, D = Dv = 512, 256 tokens: attn_x3_sfull2_kernel instead of scores + softmax.V in the latency form */
#define KEEP_ATTN_NO_SMALL (1u << 3)  /* x3, D = Dv = 64, <= 256 keys: attn_x3_kernel instead of the latency form (attn_x3_small_kernel) */

/* padding mode of keep_conv2d */
#define KEEP_UPSAMPLE_X2_PHASES 2
#define KEEP_PAD_ZERO 0
#define KEEP_PAD_REFLECT 1

int32_t keep_abi_version(void);
const char* keep_last_error(void);
/* 0 if device `dev` is a gfx950 part this library was built for, KEEP_EUNSUP otherwise */
int32_t keep_device_ok(int32_t dev);

/* ------------------------------------------------------------------------------------------------
 * keep_conv2d -- implicit-GEMM convolution / linear layer on the matrix cores.
 * Replaces torch conv2d / F.linear call sites: VQ:170-181 (ResBlock convs), VQ:135-139 (Downsample:
 * pad_t=pad_l=0, stride 2, Ho=H/2 -> the missing bottom/right taps read zeros), VQ:148-152 (Upsample:
 * `upsample`=1 folds the nearest x2 into the gather), VQ:190-217,241 (1x1 convs), KA:448-455 (CFT convs),
 * KA:78-87,145,168-169,193 (attention projections), KA:391-393,437 (MLP), GM/backbone.py (IN+ReLU convs),
 * GM/gmflow.py:45-47 (upsampler convs).
 *   out[n,oy,ox,co] = epi( bias[co] + sum_{kh,kw,ci} w[co,kh,kw,ci] * pro(in[n, oy*s-pt+kh, ox*s-pl+kw, ci]) )
 *   pro(x) = act_pro(x*pro_scale[n,ci] + pro_shift[n,ci])  (zero padding is applied AFTER pro)
 *   epi(v) = r + aux_w*(r*aux + act(v))   if aux        (CFT: dec + w*(dec*scale + shift), KA:470-471)
 *          = act(v) + r                   if residual   (ResBlock / attention residual)
 *          = act(v)                       otherwise
 * A linear layer over M tokens is N=1,H=M,W=1,KH=KW=1.  `in_ld`/`out_ld` are the pixel strides (elements)
 * of in/out (>= Cin/Cout) so slices of wider buffers can be read/written in place.
 * split_k>1: `workspace` must hold split_k*M*Cout floats; partial sums are reduced deterministically.
 */
typedef struct {
  uint32_t struct_size;   /* sizeof(keep_conv2d_args) of the CALLER's header; keep_sizeof_conv2d_args() is the library's */
  uint32_t flags;         /* KEEP_CONV_* overrides; 0 = default dispatch */
  const void* in;         /* [N,H,W,in_ld]                                   */
  const float* weight;    /* [Cout][KH][KW][Cin] fp32 (packed by the host)   */
  const float* bias;      /* [Cout] or NULL                                  */
  void* out;              /* [N,Ho,Wo,out_ld]                                */
  const float* pro_scale; /* [N,Cin] or NULL                                 */
  const float* pro_shift; /* [N,Cin] or NULL                                 */
  const void* residual;   /* [N,Ho,Wo,res_ld] or NULL                        */
  const void* aux;        /* [N,Ho,Wo,Cout] or NULL (requires residual)      */
  float* workspace;       /* split-K partials or NULL                        */
  int32_t N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int32_t in_ld, out_ld, res_ld;
  int32_t upsample; /* 1: `in` is [N,H,W,*] and is read as its nearest x2 upsampling [N,2H,2W,*];
                       KEEP_UPSAMPLE_X2_PHASES (KEEP_MMA_X3 only, v13): the same operation with the x2 folded into the WEIGHTS --
                       `weight_x3` holds four 2x2-tap phase kernels [4][Cout][3*3][Cin/16][hi16|lo16] (engine/ops.py:up2_phase_weights:
                       out[2y+py, 2x+px] only sees source rows {y-1, y} (py = 0) or {y, y+1} (py = 1), whose replicated taps are added
                       beforehand), `x3_acc_scale` is theirs; 4 of 9 taps are multiplied.  Needs H % 8 == 0, W % 32 == 0, Cout % 64 == 0,
                       no prologue / activation / aux / split-K (keep_conv2d_plan refuses otherwise).  VQ:146-156 (Upsample) */
  int32_t pro_act, epi_act;
  float aux_w;
  int32_t split_k; /* 0: the library chooses (keep_conv2d_plan reports the choice and the workspace it needs) */
  int32_t dtype; /* KEEP_F32 */
  int32_t mma;   /* KEEP_MMA_F32 | KEEP_MMA_BF16 | KEEP_MMA_X3 */
  const void* weight_bf16; /* [Cout][KH][KW][Cin] bf16, required when mma == KEEP_MMA_BF16 */
  /* optional: per-tile (sum, sumsq) of the epilogue OUTPUT per channel, [N][stats_P][Cout][2], for the next
   * GroupNorm / InstanceNorm (keep_norm_finalize with P = stats_P): saves one full read of the activation.
   * Needs split_k == 1 and Ho*Wo a multiple of the kernel's tile height (stats_P = Ho*Wo / tile height: 256 for
   * the 3x3 halo path, else 128, or 64 when Cout <= 64 or N*Ho*Wo <= 4096). */
  float* stats_out;
  int32_t stats_P;
  int32_t bk256;     /* KEEP_MMA_BF16 gather path, 64x64 tiles: K step of 256 channels with a single LDS buffer
                        (latency-bound small-M / deep-K layers); split_k then counts 256-channel steps */
  int32_t out_dtype; /* KEEP_F32, or KEEP_BF16: `out` is a bf16 tensor (projections feeding keep_attention; needs
                        Cout/out_ld %% 4 == 0, split_k == 1, no residual) */
  /* KEEP_MMA_X3: weight * 2^e split into fp16 halves, [Cout][KH*KW][Cin/16][hi x16 | lo x16] (Cin %% 16 == 0), 16-byte
   * aligned; x3_acc_scale = 2^-e (applied to the accumulators).  NULL -> the call runs on the exact-f32 kernels. */
  const void* weight_x3;
  float x3_acc_scale;
  /* KEEP_MMA_X3, inputs that are NOT normalised by the prologue (raw residual-stream tensors): per-image max |x| of the
   * input window, [N] floats from keep_absmax.  The kernels multiply image n by the power of two that puts in_amax[n]
   * just below 2^15 before splitting (and undo it on the accumulators), so no activation can leave the fp16 range and
   * small values keep a normal `lo`.  NULL: inputs are split as they are (|x| must stay below 65504). */
  const float* x3_in_amax;
  /* optional, KEEP_MMA_X3 kernels with split_k == 1 (keep_conv2d_plan: out_amax_ok): out_amax[n] = max |out[n,...]|, [N]
   * floats, zeroed and filled by this call -- the range probe of the NEXT un-normalised x3 consumer for free */
  float* x3_out_amax;
  int32_t x3_out_amax_zeroed; /* non-zero: the caller zero-filled x3_out_amax (one arena per forward pass); otherwise the
                                 library zero-fills it with a kernel of its own before the launch */
  /* K-concatenated input of a 1x1 stride-1 convolution (KEEP_MMA_X3 GEMM form only; keep_conv2d_plan rejects it elsewhere with
   * KEEP_EUNSUP): channels [0, in2_cin1) are read from `in` (row stride in_ld), channels [in2_cin1, Cin) from `in2`, a dense
   * [N*H*W, Cin - in2_cin1] tensor -- torch.cat([a, b], -1) folded into the GEMM (GM/transformer.py:182).  NULL: single input. */
  const void* in2;
  int32_t in2_cin1;
  int32_t pad_mode; /* KEEP_PAD_ZERO (default) | KEEP_PAD_REFLECT: pixels outside the (post-upsample) input mirror it like
                       nn.ReflectionPad2d(pad) (wm_facelib/parsing/parsenet.py:97,104: every ParseNet convolution); needs
                       pad_t == pad_l < min(H, W) and KEEP_MMA_F32 or KEEP_MMA_X3 */
  /* v16, optional: LayerNorm over the Cout channels of every output row, fused into the epilogue of the KEEP_MMA_X3 GEMM form:
   * out = LayerNorm(in . W^T + bias; ln_gamma, ln_beta, ln_eps) + residual -- the `merge -> norm1` and `mlp.2 -> norm2` pairs of a
   * GMFlow transformer layer (GM/transformer.py:170-187) as one launch.  Needs a 1x1 stride-1 convolution without prologue /
   * activation / aux / split-K / statistics, Cout == out_ld == 128 (one wave holds whole rows: tile <4,1,1,4>) and
   * N*Ho*Wo %% 128 == 0; keep_conv2d_plan answers KEEP_EUNSUP otherwise.  Two-pass statistics (mean, then centred squares),
   * biased variance like nn.LayerNorm.  NULL: no LayerNorm. */
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int32_t plan_ref_images; /* v18: the fixed reference batch the parity policies plan split-K / statistics partitions for (0 = 16).  A
                              deployment-wide numerics setting like `mma`: results are bit-identical across batch sizes within one value;
                              2 = latency profile for single clips */
} keep_conv2d_args;
/* smallest struct_size the library accepts: the v12 layout up to and including in2_cin1 (fields appended later are optional) */
#define KEEP_CONV2D_ARGS_V12_SIZE 256
int32_t keep_conv2d(const keep_conv2d_args* a, void* stream);
/* sizeof(keep_conv2d_args) / sizeof(keep_attention_args) as THIS library was compiled: a binding checks its own struct
 * against it at load time (engine/hiplib.py does) */
int32_t keep_sizeof_conv2d_args(void);
int32_t keep_sizeof_attention_args(void);

/* What keep_conv2d will do with exactly these arguments -- kernel choice, split-K, buffer sizes -- so that callers size
 * `workspace` / `stats_out` from the library's own decision instead of mirroring its tile rules.  Tensor pointers only
 * contribute their alignment (pass the real ones, or NULL for absent optional tensors). */
typedef struct {
  int32_t path;             /* internal kernel family id (diagnostic) */
  int32_t split_k;          /* the factor keep_conv2d will use (== args.split_k clamped, or the library's choice for 0) */
  int64_t workspace_bytes;  /* bytes `workspace` must hold (0 when split_k == 1) */
  int32_t stats_rows;       /* output pixels per statistics partial; 0: this call cannot emit `stats_out` */
  int32_t stats_P;          /* partials per image = Ho*Wo / stats_rows: stats_out is [N][stats_P][Cout][2] floats */
  int32_t wants_bf16_input; /* KEEP_MMA_BF16: faster if the caller first runs keep_norm_act_bf16 (prologue applied once,
                               bf16 tensor) and calls again with dtype = KEEP_BF16 and no prologue */
  int32_t out_bf16_ok;      /* out_dtype = KEEP_BF16 is supported for this geometry */
  int32_t out_amax_ok;      /* x3_out_amax will be filled by this call */
  char kernel[64];          /* kernel family as rocprofv3 prints it (bench.py groups its HIP-event timings by it) */
} keep_conv2d_plan_out;
int32_t keep_conv2d_plan(const keep_conv2d_args* a, keep_conv2d_plan_out* out);

/* ------------------------------------------------------------------------------------------------
 * keep_attention -- fused softmax(scale * Q K^T + mask) V  (flash style: scores never reach HBM).
 * Replaces: VQ:226-239 (AttnBlock), KA:205-234 (CrossAttention._attention, used by CFA KA:527 and the
 * Kalman blocks KA:653,679), nn.MultiheadAttention inside KA:431, GM/transformer.py:8-16,46-105 (swin
 * window attention), GM/matching.py:15-34 (global correlation soft-argmax: V = pixel grid),
 * GM/transformer.py:368-372 (flow propagation: V = flow).
 * Element offset of token t, head h, batch b of X in {q,k,v,o}:  b*x_bs + t*x_ts + h*x_hs.
 *   mode 0: plain.
 *   mode 1: sparse-causal keys (KA:704-716): batch b is frame f=b%T of a clip; key/value token t of the
 *           2*seg_len keys comes from frame 0 (t<seg_len) or frame max(f-1,0) (t>=seg_len) of the same clip.
 *   mode 2: (shifted) window attention on an img_h x img_w token grid cut into ksplit x ksplit windows
 *           (GM/transformer.py:46-105): batch = image*ksplit^2 + window, tokens are window-local, rolled by
 *           `shift`; shift>0 adds the -100 cross-region mask (GM/transformer.py:19-43).  Keys/values are read
 *           from image (image + kv_rot) % n_img  (the [f0;f1] vs [f1;f0] pairing, GM/transformer.py:301-314).
 */
typedef struct {
  uint32_t struct_size;   /* sizeof(keep_attention_args) of the caller's header */
  uint32_t flags;         /* KEEP_ATTN_* overrides; 0 = default dispatch */
  const void* q;
  const void* k;
  const void* v;
  float* o;
  int64_t q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
  int32_t B, H, Lq, Lk, D, Dv;
  float scale;
  int32_t mode;
  int32_t T, seg_len;                           /* mode 1 */
  int32_t img_h, img_w, ksplit, shift, kv_rot, n_img; /* mode 2 */
  int32_t mma; /* KEEP_MMA_F32 | KEEP_MMA_BF16 (Q,K,V,P rounded to bf16; fp32 softmax + accumulate) | KEEP_MMA_X3 (Q,K,V,P
                  split into fp16 hi + lo, three MFMAs per product: fp32-grade; exact-fp32 softmax) */
  int32_t in_dtype; /* KEEP_F32, or KEEP_BF16 (with KEEP_MMA_BF16): q, k, v are bf16 tensors, strides in elements */
  /* KEEP_MMA_X3, mode 0 only: per-batch max |q|, |k|, |v| ([B] floats each, keep_absmax) for operands that are not
   * bounded by a normalisation (CFA reads the raw residual stream); all three or none */
  const float* q_amax;
  const float* k_amax;
  const float* v_amax;
  /* optional scratch (device, 16-byte aligned) of at least keep_attention_workspace_bytes(a) bytes: with it a KEEP_MMA_X3 call
   * whose K / V are streamed by several query blocks first packs K and V^T ONCE into the split-fp16 tile images the kernel
   * keeps in LDS (keep_attn.hip: attn_pack_kv_x3_kernel) and the query blocks copy tiles instead of re-splitting them */
  void* workspace;
  int64_t workspace_bytes;
} keep_attention_args;
#define KEEP_ATTENTION_ARGS_V12_SIZE 248
int32_t keep_attention(const keep_attention_args* a, void* stream);
/* bytes of `workspace` this call can use (0: none); a smaller or NULL workspace selects the unpacked path */
int64_t keep_attention_workspace_bytes(const keep_attention_args* a);

/* ------------------------------------------------------------------------------------------------
 * Normalisation statistics.  GroupNorm(32, eps 1e-6) VQ:16-17 and InstanceNorm2d(eps 1e-5, no affine)
 * GM/backbone.py:7,41 are applied lazily: statistics are reduced here into per-(n,channel) scale/shift
 * that the consuming keep_conv2d applies in its prologue.
 *   keep_chan_stats:   part[n][p][c] = (sum, sumsq) of x[n, pixels of chunk p, c]   (P chunks per image)
 *   keep_norm_finalize: groups of C/G channels -> mean/var (biased) -> scale = gamma*rstd, shift = beta-mean*scale
 */
int32_t keep_chan_stats(const float* x, float* part, int32_t N, int32_t HW, int32_t C, int32_t ld, int32_t P,
                        void* stream);
int32_t keep_norm_finalize(const float* part, const float* gamma, const float* beta, float* scale, float* shift,
                           int32_t N, int32_t HW, int32_t C, int32_t G, int32_t P, float eps, void* stream);
/* one-launch variant for small maps: one block per (image, group) reduces H*W*(C/G) values and writes scale/shift */
int32_t keep_group_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift, int32_t N,
                         int32_t HW, int32_t C, int32_t G, float eps, void* stream);
/* out = act(x*scale[n,c]+shift[n,c]) materialised (only where no conv consumes it: GM residual join) */
int32_t keep_affine_act(const float* x, const float* scale, const float* shift, float* out, int32_t N, int32_t HW,
                        int32_t C, int32_t act, void* stream);
/* out (bf16 [N,HW,C]) = bf16( act_pro(x*scale[n,c]+shift[n,c]) ); scale/shift NULL = plain cast.  The normalise +
 * activate pass in front of the 3x3 halo convolution under KEEP_MMA_BF16 (keep_conv2d with dtype = KEEP_BF16). */
int32_t keep_norm_act_bf16(const void* x, const float* scale, const float* shift, void* out, int32_t N, int32_t HW,
                           int32_t C, int32_t act, int32_t in_dtype /* KEEP_F32 | KEEP_BF16 */, void* stream);
/* GM/transformer.py:139-142,182 (bf16 policy): out[M,C] = W2 . gelu( W0 . cat[a[M,C] | b[M,C]] ), W0 bf16 [8C,2C],
 * W2 bf16 [C,8C], no biases, C = 128; the [M,8C] intermediate stays on the CU. */
int32_t keep_gm_mlp(const float* a, const float* b, const void* w0_bf16, const void* w2_bf16, float* out, int64_t M,
                    int32_t C, void* stream);
/* v18, KEEP_MMA_X3 policy: the whole feed-forward block of a GMFlow cross-attention layer (GM/transformer.py:137-142,182-187) as one launch,
 *   out[M,C] = LayerNorm( W2 . gelu( W0 . cat[src[M,C] | msg[M,C]] ); ln_gamma, ln_beta, ln_eps ) + src,   C = 128, hidden %% 32 == 0,
 * with x3 products (three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate) and the [M, hidden] intermediate kept in registers
 * (csrc/keep_ffn_x3.hip).  w0_x3: split-fp16 twin of W0 [hidden, 2C] in the keep_conv2d `weight_x3` layout ([row][Cin/16][hi16|lo16]),
 * w0_acc_scale its 2^-e; w2p_x3: the same for W2 [C, hidden] with every group of 16 hidden units stored in the order
 * [0-3, 8-11, 4-7, 12-15] (engine/ops.py:ffn_w2_perm -- the order in which an MFMA's accumulator layout hands a lane its hidden units).
 * GELU: the x3-grade fast form (erf to 1.5e-7), or the library erff with exact_act != 0.  No biases (GMFlow's mlp has none). */
int32_t keep_gm_ffn_x3(const float* src, const float* msg, const void* w0_x3, float w0_acc_scale, const void* w2p_x3, float w2_acc_scale,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int64_t M, int32_t C, int32_t hidden,
                       int32_t exact_act, void* stream);
/* GM/transformer.py:148-176 projections at M ~ 1e6 tokens (bf16 policy): out[M,N] = x[M,128] . W[N,128]^T (+ bias),
 * W bf16, N in {128,256,384}, out fp32 or bf16 (out_dtype); persistent blocks keep W in LDS and stream the token rows. */
int32_t keep_token_linear(const float* x, const void* w_bf16, const float* bias, void* out, int64_t M, int32_t K, int32_t N,
                          int32_t out_dtype, void* stream);
/* GM/backbone.py:36: out = relu( (a*sa+ha) + relu(b*sb+hb) ); sa/ha may be NULL (identity shortcut) */
int32_t keep_gm_join(const float* a, const float* sa, const float* ha, const float* b, const float* sb,
                     const float* hb, float* out, int32_t N, int32_t HW, int32_t C, void* stream);

/* amax[n] = max |x[n, r, c]| over the R rows x C columns (row stride ld) of image n -- the range probe in front of
 * KEEP_MMA_X3 operators whose input is not normalised (x3_in_amax / q_amax ...).  Deterministic (max is order-free). */
int32_t keep_absmax(const float* x, float* amax, int32_t N, int64_t R, int32_t C, int64_t ld, int64_t img_stride,
                    int32_t zeroed /* non-zero: amax[] was zero-filled by the caller */, void* stream);

/* LayerNorm(eps 1e-5) over the last dim of [M,C] (KA:395-396,491-492,597; GM/transformer.py:134,145).
 *   y = LN(x)*gamma+beta;  out = y + (res ? res : 0);  out2 (optional) = y + pos[m % pos_rows]  (KA:429-430) */
int32_t keep_layernorm(const float* x, const float* gamma, const float* beta, const float* res, float* out,
                       const float* pos, int32_t pos_rows, float* out2, int32_t M, int32_t C, float eps,
                       void* stream);

/* GEGLU gate (diffusers FeedForward, KA:495-496,595-596): out[m,j] = x[m,j] * gelu_erf(x[m,F+j]), x is [M,2F] */
int32_t keep_geglu(const float* x, float* out, int32_t M, int32_t F, void* stream);
/* keep_layernorm (no pos / out2) and keep_geglu with the consumer's x3 range scale fused (v19): amax[n] = max |out| over the
 * rows_per_image rows of image n (M = N * rows_per_image), exactly what keep_absmax would return for the output -- the CFA block's
 * feed-forward GEMMs and the next frame's kv projection take it instead of a probe launch (keep_arch.py:519-541, engine/net.py:_cfa).
 * zeroed != 0: the caller hands in slots it has zero-filled (a per-forward arena); outputs are bit-identical to keep_layernorm / keep_geglu. */
int32_t keep_layernorm_amax(const float* x, const float* gamma, const float* beta, const float* res, float* out, int32_t M, int32_t C,
                            float eps, int32_t rows_per_image, float* amax, int32_t zeroed, void* stream);
int32_t keep_geglu_amax(const float* x, float* out, int32_t N, int32_t rows_per_image, int32_t F, float* amax, int32_t zeroed, void* stream);

/* KA:1085-1089 + VQ:78-91: idx[m] = argmax_j logits[m,j] (lowest index on ties); out[m,:] = codebook[idx[m],:].
 * force_idx (optional) overrides the argmax (parity tests).  margin (optional) = top1-top2 logit gap.
 * An arg-max is where a NaN / inf would otherwise turn into a finite, plausible, WRONG result: a row whose maximum is not
 * finite (all-NaN rows included) gets idx 0, a NaN-filled `out` row (so the failure also travels with the data) and, when
 * `status` (optional, one device int32) is given, bit KEEP_STATUS_NONFINITE_LOGITS OR-ed into it. */
#define KEEP_STATUS_NONFINITE_LOGITS 1
#define KEEP_STATUS_NONFINITE_TENSOR 2
int32_t keep_argmax_gather(const float* logits, const float* codebook, const int32_t* force_idx, int32_t* idx,
                           float* margin, float* out, int32_t M, int32_t ncodes, int32_t dim, int32_t* status, void* stream);
/* status |= KEEP_STATUS_NONFINITE_TENSOR if any of the n floats of x is NaN or +-inf (one pass, one atomic per block at
 * most): the x3 policy's end-of-forward range check as a 4-byte read instead of a host-side reduction */
int32_t keep_nonfinite_flag(const float* x, int64_t n, int32_t* status, void* stream);

/* VQ:37-48 true nearest-neighbour code search: idx[m] = argmin_j |z_m|^2 + |e_j|^2 - 2 z_m.e_j  (next-row 8f-3) */
int32_t keep_vq_nearest(const float* z, const float* codebook, int32_t* idx, int32_t M, int32_t ncodes, int32_t dim,
                        void* stream);

/* KA:796-799: z_hat = (1-g)*z_code + g*z_prime, g [N,HW] broadcast over C */
int32_t keep_kalman_update(const float* z_code, const float* z_prime, const float* gain, float* out, int32_t N,
                           int32_t HW, int32_t C, void* stream);

/* AU:113-144 flow_warp: bilinear grid_sample(zeros, align_corners=True) of x [N,H,W,C] by flow [N,H,W,2].  A non-finite
 * flow vector yields NaN pixels (torch's grid_sample does the same for NaN; an out-of-range sample must not hide it). */
int32_t keep_flow_warp(const float* x, const float* flow, float* out, int32_t N, int32_t H, int32_t W, int32_t C,
                       void* stream);

/* GM/gmflow.py:75-88 convex upsampling: mask [N,H,W,9*k*k] (channel = tap*k*k + ky*k + kx), flow [N,H,W,2]
 * -> out [N,kH,kW,2] */
int32_t keep_convex_upsample(const float* mask, const float* flow, float* out, int32_t N, int32_t H, int32_t W,
                             int32_t k, void* stream);

/* KA:1020-1023 (need_upscale): F.interpolate(scale_factor=scale, mode='bilinear', align_corners=False) of `planes` planar
 * H x W images (a [B*T,3,H,W] clip is planes = B*T*3) -> [planes, scale*H, scale*W] */
int32_t keep_bilinear_upscale(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t scale, void* stream);

/* layout / elementwise helpers */
/* [N,C,H,W] -> [N,H,W,C];  mode 1 additionally applies GMFlow's input normalisation (GF:56-57, GM/utils.py:55-63) */
int32_t keep_nchw_to_nhwc(const float* x, float* out, int32_t N, int32_t C, int32_t HW, int32_t mode, void* stream);
/* v17: x [N,3,H,W] in [-1,1] -> out [N,H/2,W/2,16]: GMFlow's input normalisation (GF:56-57, GM/utils.py:55-63, the op order of mode 1
 * above) and a 2x2 space-to-depth in one pass; channel (dy*2 + dx)*3 + c = pixel (2Y+dy, 2X+dx) of colour c, channels 12..15 zero.
 * The 7x7 stride-2 pad-3 first convolution of GMFlow's encoder (GM/backbone.py:69) is then a 4x4 stride-1 convolution with
 * pad_t = pad_l = 2 on this image (weights repacked by the host: tap ky of the 7x7 kernel = s2d tap (ky + 1) / 2, sub-row
 * (ky + 1) & 1), which keep_conv2d runs on its 16-channel MFMA kernels.  H, W even. */
int32_t keep_rgb_s2d(const float* x, float* out, int32_t N, int32_t H, int32_t W, void* stream);
int32_t keep_nhwc_to_nchw(const float* x, float* out, int32_t N, int32_t C, int32_t HW, void* stream);
/* out[n,i] = a[n,i] + alpha * t[i % tsize]   (position tables, grid subtraction) */
int32_t keep_add_bcast(const float* a, const float* t, float* out, int64_t total, int64_t tsize, float alpha,
                       void* stream);
/* out[m, 0:C1] = a[m,:], out[m, C1:C1+C2] = b[m,:], out[m, C1+C2:out_ld] = 0 (channel padding to a multiple of 16) */
int32_t keep_concat2(const float* a, const float* b, float* out, int64_t M, int32_t C1, int32_t C2, int32_t out_ld, void* stream);
/* img_util.py:66-90 tensor2img: fp32 [N,H,W,3] RGB -> uint8 [N,H,W,3] BGR, clamp[-1,1], round-half-even */
int32_t keep_tensor2img(const float* x, uint8_t* out, int64_t npix, void* stream);
/* keep_processor.py:258-259: uint8 BGR [N,H,W,3] -> fp32 NHWC RGB (float32(u8/255.) - 0.5)/0.5 */
int32_t keep_img2tensor(const uint8_t* x, float* out, int64_t npix, void* stream);
/* v18: modules/utils.py:cv2_to_comfy_image (reference utils.py:162-166) on the device: uint8 BGR [N,H,W,3] -> the ComfyUI IMAGE layout,
 * fp32 RGB in [0,1] = float32(u8) / 255 (one correctly rounded division, bit-equal to numpy's) */
int32_t keep_bgr_u8_to_comfy(const uint8_t* x, float* out, int64_t npix, void* stream);
/* v20: modules/utils.py:comfy_image_to_cv2 (reference utils.py:155-160) on the device: the ComfyUI IMAGE layout fp32 RGB [N,H,W,3] ->
 * uint8 BGR, `(x * 255).astype(np.uint8)`: one float32 multiply, truncation towards zero through a 32-bit integer whose low byte is
 * kept (numpy on x86-64: 256.0 -> 0, -1.0 -> 255; NaN / inf / beyond int32 -> 0).  x 16-byte aligned, out 4-byte aligned. */
int32_t keep_comfy_to_bgr_u8(const float* x, uint8_t* out, int64_t npix, void* stream);


/* ---- face parsing (wm_facelib/parsing/parsenet.py on keep_conv2d with KEEP_PAD_REFLECT, engine/parsenet.py) ----
 * out[m] = argmax over the first C channels of row m of x [M, ld] (lowest index on ties): ParseNet's out.argmax(dim=1),
 * face_restoration_helper.py:424, on channels-last logits. */
int32_t keep_channel_argmax(const float* x, uint8_t* out, int64_t M, int32_t C, int32_t ld, void* stream);

/* ---- face detection (wm_facelib/detection/retinaface on keep_conv2d, engine/retinaface.py) ----
 * nn.MaxPool2d(3, stride 2, padding 1) of the ResNet-50 stem on an NHWC map: [N,H,W,C] -> [N,(H-1)/2+1,(W-1)/2+1,C] */
int32_t keep_maxpool3s2(const float* x, float* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* depthwise 3x3 convolution, padding 1, stride 1 or 2, + bias + activation (v15): the first half of a conv_dw block of the
 * retinaface_mobile0.25 trunk with its BatchNorm folded (retinaface_net.py:25-34: Conv2d(inp, inp, 3, stride, 1, groups=inp) +
 * BatchNorm2d + LeakyReLU(0.1)).  x [N,H,W,C] NHWC, w [3][3][C] (tap-major: w[ky][kx][c] = weight[c, 0, ky, kx]), bias [C] or
 * NULL, out [N,(H-1)/stride+1,(W-1)/stride+1,C]; C % 4 == 0.  Taps are accumulated ky-major, kx-minor in float32. */
int32_t keep_dwconv3x3(const float* x, const float* w, const float* bias, float* out, int32_t N, int32_t H, int32_t W, int32_t C,
                       int32_t stride, int32_t act, void* stream);
/* RetinaFace.detect_faces, ordering + NMS of the survivors on the device (v15; retinaface.py:240-246, retinaface_utils.py:39-47 =
 * torchvision.ops.nms): dets / counts as keep_retina_decode left them -> out [N, cap, 16]: the kept rows of each frame in descending
 * score order (equal scores: descending anchor index), out_counts[n] their number, or -1 when counts[n] > cap (host path).
 * cap <= 4096.  float32 IoU: inter / (area_i + area_j - inter) > iou_threshold, areas (x2 - x1) * (y2 - y1). */
int32_t keep_retina_nms(const float* dets, const int32_t* counts, float* out, int32_t* out_counts, int32_t N, int32_t cap,
                        float iou_threshold, void* stream);
/* The same suppression and compaction for frames whose ORDER the caller made (v19): order [N, cap] int32, order[n, i] = the row of dets[n]
 * of rank i (0 .. counts[n]) -- the frames keep_retina_nms hands back because two survivors share a score (out_counts = -2): the host
 * orders them as the reference does (`scores.argsort()[::-1]`, numpy's own tie order) and only the permutation goes up. */
int32_t keep_retina_nms_ordered(const float* dets, const int32_t* counts, const int32_t* order, float* out, int32_t* out_counts, int32_t N,
                                int32_t cap, float iou_threshold, void* stream);
/* ---- YOLOv5-face detectors on keep_conv2d (v15; wm_facelib/detection/yolov5face/models/common.py, yolo.py; engine/yoloface.py) ----
 * nn.MaxPool2d(k, stride, padding = pad, ceil_mode) on a channel slice of an NHWC map (padding = -inf): StemBlock's 2x2 stride-2
 * ceil-mode pool (common.py:53) and SPP's k x k stride-1 pools (common.py:160-163).  x rows of in_ld floats, out rows of out_ld
 * floats (the pointers already point at the slices' first channels), C % 4 == 0; Ho / Wo: the output size the caller computed. */
int32_t keep_maxpool2d(const float* x, float* out, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_ld, int32_t out_ld,
                       int32_t k, int32_t stride, int32_t pad, int32_t Ho, int32_t Wo, void* stream);
/* channel-slice copy with optional nearest x2 upsampling: dst[n, y, x, 0..C) = src[n, y >> up, x >> up, 0..C) -- torch.cat along
 * channels (common.py Concat) one source at a time, nn.Upsample(None, 2, 'nearest') folded in.  H, W: the DESTINATION size. */
int32_t keep_slice_copy(const float* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t src_ld, int32_t dst_ld,
                        int32_t up, void* stream);
/* ShuffleV2Block tail (common.py:148-155): channel_shuffle(cat(a, b), 2) -- out[r, 2 i] = a[r, i], out[r, 2 i + 1] = b[r, i];
 * a / b rows of a_ld / b_ld floats (a is usually the untouched first half of the block's input), out dense [rows, 2 half]. */
int32_t keep_channel_shuffle2(const float* a, const float* b, float* out, int64_t rows, int32_t half, int32_t a_ld, int32_t b_ld,
                              void* stream);
/* Detect.forward, inference branch (yolo.py:44-78) for one level: raw [N, ny, nx, 3 * 16] (the level's 1x1 convolution, NHWC =
 * the reference's permute(0, 1, 3, 4, 2) up to the anchor axis) -> pred[n, row0 + (a * ny + y) * nx + x, 0..16): sigmoid on box /
 * objectness / class, grid + anchor decode of the box and of the five landmarks, in pixels of the network input. */
int32_t keep_yolo_decode(const float* raw, float* pred, int32_t N, int32_t ny, int32_t nx, float stride, const float* anchors_wh,
                         int32_t row0, int32_t rows_total, void* stream);
/* YoloDetector._preprocess on the device (v19; face_detector.py:48-67, utils/datasets.py:5-36): frames uint8 [N,H,W,3] ->
 * out float32 [N,H2,W2,3] NHWC in [0, 1]: optional channel swap (cv2.cvtColor BGR2RGB, face_detector.py:126), cv2.resize(INTER_LINEAR) to
 * (rw, rh) when that differs from (W, H) -- OpenCV's 8-bit fixed-point bilinear (11-bit coefficients; an exact 2x reduction, which cv2
 * turns into INTER_AREA, is refused), placed at (top, left) of a canvas of 114 (cv2.copyMakeBorder), then float / 255. */
int32_t keep_yolo_letterbox_u8(const uint8_t* frames, float* out, int32_t N, int32_t H, int32_t W, int32_t rh, int32_t rw, int32_t top,
                               int32_t left, int32_t H2, int32_t W2, int32_t swap_rb, void* stream);
/* non_max_suppression_face up to its NMS call (v19; utils/general.py:89-143, one class): pred [N,P,16] as keep_yolo_decode wrote it ->
 * for every row with objectness > conf_threshold and conf = class * objectness > conf_threshold one row of dets [N,cap,16] =
 * x1 y1 x2 y2 conf lm0..lm9 row-index (xywh2xyxy: x -+ w / 2), appended in arrival order; counts[n] (zeroed by the caller) = survivors
 * of frame n (may exceed cap: rows beyond cap are dropped).  keep_retina_nms (= torchvision.ops.nms) takes dets / counts from here. */
int32_t keep_yolo_select(const float* pred, float* dets, int32_t* counts, int32_t N, int32_t P, int32_t cap, float conf_threshold,
                         void* stream);
/* FPN top-down step (retinaface_net.py:86-92): out = a + nearest-resize(b [N,hb,wb,C] -> [N,H,W,C]) */
int32_t keep_upsample_add(const float* a, const float* b, float* out, int32_t N, int32_t H, int32_t W, int32_t hb, int32_t wb,
                          int32_t C, void* stream);
/* x = act(x) in place, act = KEEP_ACT_*: the ReLU after a Bottleneck's residual sum (torchvision resnet.py Bottleneck.forward) */
int32_t keep_act_inplace(float* x, int64_t n, int32_t act, void* stream);
/* RetinaFace.detect_faces post-processing (retinaface.py:231-246; decode / decode_landm: retinaface_utils.py:254-294) on the
 * device: heads [N,P,32] = per pixel [cls a0 a1 | box a0 a1 | landmarks a0 a1] of the fused head convolutions, priors [2P,4]
 * (cx,cy,w,h).  For every anchor with softmax(cls)[1] > conf_threshold one row of dets [N,cap,16] = x1 y1 x2 y2 score lm0..lm9
 * anchor-index (scaled to pixels by scale_x / scale_y), appended in arrival order; counts[n] (zeroed by the caller) = survivors of
 * frame n (may exceed cap: rows beyond cap are dropped, the caller re-runs its host decoder).  ABI v14. */
int32_t keep_retina_decode(const float* heads, const float* priors, float* dets, int32_t* counts, int32_t N, int32_t P, int32_t cap,
                           float var0, float var1, float scale_x, float scale_y, float conf_threshold, void* stream);

/* ---- paste-back compositing (face_restoration_helper.py:346-475, use_parse=True branch) ----------------
 * Separable filter with BORDER_REFLECT_101 (cv2.GaussianBlur, :433-434): n images [H,W]; the input is `src` (float) or a
 * class map `classes` (uint8) looked up through `lut` (MASK_COLORMAP, :428-429).  tmp and dst: [n,H,W] float.  kern: device
 * pointer to ntap (odd, <= 128) float taps (cv2.getGaussianKernel). */
int32_t keep_sep_filter(const float* src, const uint8_t* classes, const float* lut, float* tmp, float* dst, int32_t n, int32_t H,
                        int32_t W, const float* kern, int32_t ntap, void* stream);
/* uint8 -> float32 (the frame accumulator of :463) and back: clip [0,255], round half to even, uint8 (:465-468) */
int32_t keep_u8_to_f32(const uint8_t* x, float* out, int64_t n, void* stream);
int32_t keep_f32_round_u8(const float* x, uint8_t* out, int64_t n, void* stream);
/* cv2.warpAffine(src uint8 [H,W,3], M, (dw, dh)) with INTER_LINEAR, BORDER_CONSTANT and a border colour: the similarity crop that
 * PRODUCES the aligned faces (align_warp_face, face_restoration_helper.py:316-318: borderValue (135, 133, 132)).  dst_to_src:
 * HOST pointer to the 6 doubles of the destination -> source map (cv2.invertAffineTransform of M), read at call time. */
int32_t keep_warp_affine_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t dh, int32_t dw, const double* dst_to_src,
                            int32_t border_b, int32_t border_g, int32_t border_r, void* stream);
/* use_parse=False soft mask (FH:386-415): keep_warp_ones = cv2.warpAffine(np.ones((fh, fw), float32), M, (W, H)) for the whole
 * frame; keep_erode_rect = cv2.erode(img, np.ones((k, k))) (separable minimum, +inf outside the image), tmp / dst: [H,W] floats. */
int32_t keep_warp_ones(float* dst, int32_t H, int32_t W, int32_t fh, int32_t fw, const double* dst_to_src, void* stream);
/* draw_box (v15; FH:393-400,467-475): the warped border mask of one face -- ones(fh, fw) with cv2.rectangle((t, t), (fw - t - 1,
 * fh - t - 1), 0, filled) -- thresholded at 0.5 and painted (0, 255, 0) into the rounded uint8 frame [H,W,3], inside the face's
 * bounding box [x0, x1) x [y0, y1). */
int32_t keep_draw_box(uint8_t* frame, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t thickness, const double* dst_to_src,
                      int32_t x0, int32_t y0, int32_t x1, int32_t y1, void* stream);
int32_t keep_erode_rect(const float* src, float* tmp, float* dst, int32_t H, int32_t W, int32_t k, void* stream);
/* One face into the float frame [H,W,3], in place, over the box [x0,x1) x [y0,y1): cv2.warpAffine(face uint8 [fh,fw,3]) (:382)
 * and cv2.warpAffine(mask float [fh,fw]) (:441) with INTER_LINEAR / BORDER_CONSTANT 0 and OpenCV's fixed-point coordinates,
 * the mask's `mask_border` outer rows / columns read as zero and its values divided by 255 (:435-437), then
 * frame = soft * face + (1 - soft) * frame in float32 (:463).  dst_to_src: HOST pointer to the 6 doubles of the
 * destination -> source map (cv2.invertAffineTransform of the matrix the reference passes), read at call time.
 * mask_border < 0: `mask` is a soft mask already in FRAME space [H,W] (the use_parse=False path), read at the destination pixel. */
int32_t keep_paste_face(float* frame, int32_t H, int32_t W, const uint8_t* face, const float* mask, int32_t fh, int32_t fw,
                        const double* dst_to_src, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t mask_border, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KEEP_HIP_H */
