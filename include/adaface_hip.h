/* adaface_hip.h — C ABI of libadaface_hip.so, the MI355X (gfx950) denoising path.
 *
 * The reference (zyt334/AdaFace) has NO FFI / plugin ABI: its seam is Python — yaml
 * `target:` strings resolved by ldm/util.py:105-112 (instantiate_from_config) and
 * duck-typed calls from scripts/stable_txt2img.py:701-715.  This header is therefore
 * the contract the Python classes in adaface_amd/ldm/ bind with ctypes; each entry
 * point cites the reference function it replaces.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; af_last_error() has the text
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller (PyTorch);
 *     the library never frees or retains them past the call (stream-ordered)
 *   - tensors crossing the boundary use the REFERENCE's layout and dtype
 *     (NCHW float32, int64 timesteps); NHWC bf16/f32 is internal
 *   - `stream` is a hipStream_t (0 = default stream); a HANDLE is not thread-safe (one host thread at a time per handle), but
 *     distinct handles may be driven from distinct host threads: the library's process-wide state (launch counters, the
 *     per-kernel LDS attribute masks, the last plan) is atomic / locked; the counters then count all threads' launches
 */
#ifndef ADAFACE_HIP_H
#define ADAFACE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct af_handle af_handle;

enum { AF_DTYPE_BF16 = 0, AF_DTYPE_F32 = 1 };
enum {
  AF_OK = 0,
  AF_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  AF_ERR_HIP = -2,       /* HIP runtime error */
  AF_ERR_NAME = -3,      /* unknown tensor name */
  AF_ERR_STATE = -4      /* call sequence error (weights/context missing) */
};

/* Mirrors the constructor kwargs of UNetModel (ldm/modules/diffusionmodules/openaimodel.py:447-473)
 * and Decoder / AutoencoderKL (ldm/modules/diffusionmodules/model.py:502-507,
 * ldm/models/autoencoder.py:286-300) as used by configs/stable-diffusion/v1-inference-ada.yaml:35-76. */
typedef struct af_config {
  int dtype;                      /* AF_DTYPE_BF16 (throughput) or AF_DTYPE_F32 (parity) */
  /* UNet */
  int build_unet;
  int in_channels, model_channels, out_channels, num_res_blocks;
  int n_attention_resolutions, attention_resolutions[8];
  int n_channel_mult, channel_mult[8];
  int num_heads, context_dim, transformer_depth;
  int n_context_layers;           /* 16 = AdaFace layerwise context, openaimodel.py:863-883 */
  /* VAE decoder */
  int build_vae;
  int vae_ch, vae_out_ch, vae_num_res_blocks, vae_z_channels, vae_embed_dim;
  int n_vae_ch_mult, vae_ch_mult[8];
  /* VAE encoder (init-image path: Encoder model.py:408-499 + quant_conv autoencoder.py:304); needs build_vae */
  int build_vae_encoder;
  int vae_in_channels;            /* 3 */
  /* CLIP text tower = FrozenCLIPEmbedder.transformer (ldm/modules/encoders/modules.py:179-463; transformers
   * CLIPTextModel: openai/clip-vit-large-patch14 = vocab 49408, hidden 768, 12 layers, 12 heads, intermediate 3072,
   * 77 positions, quick_gelu) — the conditioning producer in front of the path */
  int build_clip;
  int clip_vocab, clip_hidden, clip_layers, clip_heads, clip_intermediate, clip_max_pos;
} af_config;

const char* af_last_error(void);
int af_version(void);

/* LatentDiffusion.__init__ model assembly + model.to(device)
 * (ldm/models/diffusion/ddpm.py:714-813, scripts/stable_txt2img.py:428). */
int af_create(int device_id, const af_config* cfg, af_handle** out);
void af_destroy(af_handle* h);

/* model.load_state_dict (ldm/util.py:129): one call per state_dict entry, fp32 HOST data.
 * `name` is the checkpoint key: "model.diffusion_model.<k>" or "first_stage_model.<k>". */
int af_load_tensor(af_handle* h, const char* name, const float* host_data, int ndim, const int64_t* shape);
/* same, from fp32 DEVICE data (synthetic weights generated on the GPU; no host round trip) */
int af_load_tensor_device(af_handle* h, const char* name, const float* dev_data, int ndim, const int64_t* shape);
int af_num_tensors(af_handle* h);
const char* af_tensor_name(af_handle* h, int i);
int af_tensor_loaded(af_handle* h, int i);
int af_tensor_shape(af_handle* h, int i, int64_t* shape4 /* up to 4 dims, 0-terminated */);

/* Subject-token convolutional attention inside the UNet's cross-attention layers
 * (extra_info['use_conv_attn_kernel_size'] / ['placeholder2indices'], openaimodel.py:852-853,922-945;
 * CrossAttention.forward attention.py:208-216; replace_rows_by_conv_attn ldm/util.py:701-879).
 * ks = 2, 3 or 4 (ldm/util.py:747-760; <= 1 or n_subj = 0 switches it off); batch_idx[n_subj] = samples of the CFG batch
 * that carry the subject, token_idx[n_subj][ks*ks] = text positions of its first ks*ks embeddings in tap order (host
 * arrays).  Applies to every
 * conditioned layer except CA layers 6-10, as the reference.  Must be followed by af_set_context. */
int af_set_conv_attn(af_handle* h, int ks, int n_subj, const int* batch_idx, const int* token_idx);
/* get_layer_context + to_k/to_v of all cross-attention layers, hoisted out of the
 * step loop (openaimodel.py:863-920, attention.py:195-196).  ctx_dev: fp32
 * [Bf*n_layers, n_tokens, context_dim] laid out as the reference does
 * (layer index inside the batch axis, embedding_manager.py:1342-1353), or
 * [Bf, n_tokens, context_dim] when layerwise == 0. */
int af_set_context(af_handle* h, const float* ctx_dev, int Bf, int n_tokens, int layerwise, void* stream);

/* UNetModel.forward (openaimodel.py:827-1052): x_dev [Bf,Cin,H,W] fp32 NCHW,
 * t_dev [Bf] int64, eps_dev [Bf,Cout,H,W] fp32 NCHW.  Uses the context set above. */
int af_unet_forward(af_handle* h, const float* x_dev, const int64_t* t_dev, float* eps_dev, int Bf, int H, int W,
                    void* stream);
/* The same forward on the classifier-free-guidance batch [x; x], [t; t] that p_sample_ddim / p_sample_plms build with
 * torch.cat([x] * 2) (ddim.py:236-247, plms.py:181-192): x_dev [Bf/2,Cin,H,W], t_dev [Bf/2], eps_dev [Bf,Cout,H,W]
 * (first half = the first Bf/2 contexts of af_set_context, i.e. cond first as the reference), Bf even.  Everything in
 * front of the first cross-attention (time embedding, conv_in, input_blocks[1]'s ResBlock, the first transformer's
 * GroupNorm / proj_in / self-attention) is identical for the two halves, so it is computed for Bf/2 samples and copied.
 * Same result as af_unet_forward on the concatenated inputs up to the summation order of shape-dependent kernel plans. */
int af_unet_forward_twin(af_handle* h, const float* x_dev, const int64_t* t_dev, float* eps_dev, int Bf, int H, int W,
                         void* stream);

/* ---- conditioning producer: CLIP text tower (names "cond_stage_model.transformer.text_model.<k>") ----
 * af_clip_embed_tokens = CLIPTextEmbeddings.token_embedding (encoders/modules.py:207-208): ids_dev [n] int64 ->
 *   emb_dev [n, hidden] fp32.  The caller (EmbeddingManager.forward, embedding_manager.py:1292-1584) patches the
 *   placeholder rows and tucks the 16 layer copies into the batch axis before the encoder runs.
 * af_clip_text_forward = the rest of text_model_forward (encoders/modules.py:299-371): + position embeddings, the
 *   causally masked pre-LN transformer layers, the weighted sum of the LAST TWO hidden states (w_prev for the input of
 *   the last layer, w_last for its output; the reference's default is 0.5 / 0.5), final_layer_norm.
 *   inputs_embeds_dev [Bn, T, hidden] fp32 -> out_dev [Bn, T, hidden] fp32. */
int af_clip_embed_tokens(af_handle* h, const int64_t* ids_dev, int64_t n, float* emb_dev, void* stream);
int af_clip_text_forward(af_handle* h, const float* inputs_embeds_dev, int Bn, int T, float w_prev, float w_last,
                         float* out_dev, void* stream);
/* the same with THREE blended hidden states (w_prev2 for the input of the second-to-last layer): CLIPTextModelWrapper.forward
 * with hidden_state_layer_weights (ldm/modules/arc2face_models.py:230-243), the zero-shot identity path of SURVEY.md 8f-4
 * (SubjBasisGenerator.prompt2token_proj, weights [1, 2, 4] / 7).  w_prev2 = w_prev = 0, w_last = 1: plain last state. */
int af_clip_text_forward3(af_handle* h, const float* inputs_embeds_dev, int Bn, int T, float w_prev2, float w_prev, float w_last,
                          float* out_dev, void* stream);

/* Diagnostic tap on the U-Net's block outputs (what a forward hook on input_blocks[i] / middle_block / output_blocks[j]
 * of the reference UNetModel sees, openaimodel.py:984-1027): blocks are numbered in forward order, input_blocks
 * 0..n_in-1, middle_block = n_in, output_blocks = n_in+1+j.  af_unet_block_shape gives the (C, H, W) of a block's
 * output for an H x W latent; af_unet_set_tap makes every following af_unet_forward also write that block's output as
 * fp32 NCHW [Bf, C, H, W] to out_dev (block < 0 or out_dev NULL: off).  Used by the parity tests to localise a
 * deviation; costs nothing when off. */
int af_unet_num_blocks(af_handle* h);
int af_unet_block_shape(af_handle* h, int block, int H, int W, int* C_out, int* H_out, int* W_out);
int af_unet_set_tap(af_handle* h, int block, float* out_dev);

/* p_sample_ddim's CFG combine + x_{t-1} update (ddim.py:260,273-295), fp32, n elements.
 * eps_uncond_dev / noise_dev / pred_x0_dev may be NULL. */
int af_ddim_step(const float* x_dev, const float* eps_cond_dev, const float* eps_uncond_dev, const float* noise_dev,
                 int64_t n, float guidance, float a_t, float a_prev, float sqrt_one_minus_at, float sigma_t,
                 float temperature, float* x_prev_dev, float* pred_x0_dev, void* stream);

/* out = w0 x0 + w1 x1 + w2 x2 + w3 x3 (NULL inputs skipped), fp32, n elements; mode 1: out = x1 + w0 (x0 - x1) = the CFG
 * combine.  PLMS's Adams-Bashforth mixes of noise predictions (ldm/models/diffusion/plms.py:199,236-249). */
int af_lincomb(float* out_dev, int64_t n, const float* x0_dev, float w0, const float* x1_dev, float w1,
               const float* x2_dev, float w2, const float* x3_dev, float w3, int mode, void* stream);

/* decode_first_stage + AutoencoderKL.decode (ddpm.py:1251-1308, autoencoder.py:330-333):
 * z_dev [B,zc,H,W] fp32 -> img_dev [B,out_ch,8H,8W] fp32 NCHW (may be NULL) and/or
 * u8_dev [B,8H,8W,3] uint8 HWC = clamp((x+1)/2,0,1)*255 truncated (stable_txt2img.py:715,764-765). */
int af_vae_decode(af_handle* h, const float* z_dev, float scale_factor, float* img_dev, uint8_t* u8_dev, int B, int H,
                  int W, void* stream);

/* clamp((x+1)/2,0,1)*255 -> uint8 HWC from an fp32 NCHW image [B,3,H,W]. */
/* AutoencoderKL.encode up to the posterior parameters (autoencoder.py:324-326): Encoder.forward + quant_conv.
 * x_dev NCHW [B, in_channels, H, W] fp32 (H, W multiples of 2^(levels-1)); moments_dev NCHW
 * [B, 2*embed_dim, H/f, W/f] fp32 = (mean | logvar), f = 2^(levels-1). */
int af_vae_encode(af_handle* h, const float* x_dev, float* moments_dev, int B, int H, int W, void* stream);
/* DiagonalGaussianDistribution.sample / .mode (distributions.py:24-37,61-62) + get_first_stage_encoding
 * (ddpm.py:947-954): z = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise); noise_dev null = mode().
 * moments [B, 2C, HW], noise / z [B, C, HW], all NCHW fp32. */
int af_posterior_sample(const float* moments_dev, const float* noise_dev, float scale, float* z_dev, int B, int C,
                        int HW, void* stream);
int af_to_uint8(const float* img_dev, uint8_t* u8_dev, int B, int H, int W, void* stream);

/* bytes of the activation arena currently reserved by the handle (diagnostics) */
int64_t af_arena_bytes(af_handle* h);

/* ---- per-kernel-class HIP-event timing (bench.py roofline leg) ----
 * classes: 0 conv_gemm (implicit-GEMM conv/linear on the four-wave / halo kernels), 1 attention, 2 groupnorm,
 * 3 layernorm, 4 other, 5 conv_gemm_pp_kernel<160,gather> (3x3 / strided convs on the eight-wave ping-pong kernel),
 * 6 conv_gemm_pp_kernel<160,plain> (1x1 convs / linears), 7 conv_gemm_pp_kernel<128,*> (GEGLU, VAE widths),
 * 8 conv_gemm_pp_kernel<*,*,0,true> (fp8 operands, af_set_fp8), 9 conv3x3_halo8_kernel (3x3 / stride-1 convolutions with an
 * LDS-resident input halo: the largest single kernel of a bf16 step).
 * While enabled every launch of a class is bracketed by hipEventRecord on ITS stream; af_prof_collect
 * sums elapsed ms, launch counts and the ALGORITHMIC flops / bytes of those launches per class. */
int af_prof_enable(int class_mask); /* bit c set = time class c; 0 = off */
int af_prof_reset(void);
/* time only every `every`-th launch of a class (default 1 = all).  An event pair costs ~9 us of stream time, so
 * bench.py samples (every = 7) inside its timed region; sums and launch counts then cover the sampled launches. */
int af_prof_set_stride(int every);
/* FLOPs (2 x MAC) handed to the GEMM / convolution / attention kernels since the last reset: what the path EXECUTES
 * (bench.py: whole_path_executed_flops_frac; the phase-decomposed upsamplers and the shared CFG prefix execute fewer than
 * the reference's algorithm counts).  reset != 0 zeroes the counter after reading. */
double af_flops_issued(int reset);
/* Device clock probe: a register-only bf16 MFMA loop (`iters` rounds of four v_mfma_f32_32x32x16_bf16 per wave, two waves
 * per SIMD on every CU; <= 0: 40000 rounds, about 5 ms) stamped with s_memtime / s_memrealtime.  mfma_mhz = median in-kernel
 * clock the chip held under it, mfma_tflops = its rate.  bench.py stamps each line with both so that lines measured on
 * different devices of a pool can be compared. */
int af_clock_probe(void* stream, int iters, double* mfma_mhz, double* mfma_tflops);
int af_prof_collect(int n_classes, double* ms, int64_t* launches, double* flops, double* bytes);
/* microseconds an EMPTY event pair measures on `stream` (mean of n): the bracket's own cost inside every timed launch */
double af_prof_event_overhead_us(void* stream, int n);
/* diagnostics: the tiling the most recent conv / linear launch of this process used.
 * tile: 0-3 = 128x128 / 64x128 / 128x64 / 64x64 four-wave tiles, 4 / 5 = 256x128 / 256x160 eight-wave ping-pong tiles;
 * halo_tw != 0: LDS-halo 3x3 kernel.  The parity tests use it to assert which kernel they exercised. */
int af_last_gemm_plan(int* tile, int* splitk, int* halo_tw);
/* launches per tiling since the last reset: counts10[0..5] by tile (as af_last_gemm_plan), [6] LDS-halo 3x3 kernel,
 * [7] launches that sliced K (also counted under their tile), [8] / [9] ping-pong launches whose epilogue applied a
 * folded LayerNorm / produced LayerNorm row statistics.  Lets a whole-model test assert which kernels it ran. */
int af_gemm_plan_counts(int64_t* counts10);
int af_gemm_plan_counts_reset(void);

/* ---- tuning / diagnostic knobs ----
 * The planner thresholds and "force this kernel variant" switches live in one struct that is filled once from the
 * AF_* environment variables when the library is loaded (AF_GEMM_PP_MINFILL -> "gemm_pp_minfill", ...); nothing on the
 * launch path reads the environment.  The parity tests use af_knob_set to reach a kernel variant regardless of the
 * planner's choice and af_knob_reset to restore the load-time values.  No knob changes results beyond the summation
 * order of the chosen tiling.  The 26 names (adaface_amd/csrc/af_common.h, struct AfKnobs): splitk_target, conv_halo, gemm_pp,
 * gemm_pp_minfill, gemm_tile, gemm_splitk, gemm_groupm, gemm_dma, pp_direct, attn_ring, gn_small,
 * conv_tap_inner, ln_fuse, geglu_rowpanel, conv_halo8, conv_fast_taps, pp_stagger, gn_producer, conv_up_phase4, pp_sched,
 * attn_short, gemm_m128, small_m_tile64, gn_consumer, xattn_fused, plan_log.  Round 4 removed the six that selected a measured-neutral or
 * slower variant or nothing at all (gn_reduce, splitk_inlaunch, rowpanel_deep, gn_fold, attn_w4, gemm_pp_geglu_minkt; numbers in DESIGN.md section 5). */
int af_knob_set(const char* name, int value);
int af_knob_get(const char* name, int* value);
int af_knob_reset(void);

/* ---- operator-level entry points (parity tests; reference layouts, fp32 device tensors) ----
 * Each converts to the internal NHWC `dtype` layout, runs the same kernel the model
 * uses, and converts back. */
/* F.conv2d(x, w, b, stride, padding) with optional nearest-2x upsample of x first;
 * w [Cout,Cin,k,k] (k = 1 or 3), residual / out NCHW [B,Cout,Ho,Wo]. */
int af_op_conv2d(int dtype, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                 float* y_dev, int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int upsample,
                 void* stream);
/* F.linear on [M,K] rows: y = x w^T + b (+ residual); geglu != 0: w is [2*Nout,K] and
 * y [M,Nout] = value * gelu(gate)  (attention.py:32-45). */
int af_op_linear(int dtype, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                 float* y_dev, int64_t M, int K, int N, int geglu, void* stream);
/* F.group_norm(x, 32, gamma, beta, eps) on NCHW, optional SiLU. */
/* conv3x3 (stride 1, bf16) + GroupNorm(32)(+SiLU) with the GroupNorm statistics summed in the convolution's epilogue, as the
 * ResBlocks run the pair (openaimodel.py:259-279): h_dev = convolution output, y_dev = GroupNorm output, both fp32 NCHW */
int af_op_conv_gn(const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                  const float* gamma_dev, const float* beta_dev, float eps, int silu, float* h_dev, float* y_dev, int B,
                  int Cin, int H, int W, int Cout, void* stream);
int af_op_groupnorm(int dtype, const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, int silu,
                    float* y_dev, int B, int C, int H, int W, void* stream);
/* fp8 (OCP e4m3) operand variant of the UNet's ResBlock convolutions (BASELINE config 4: "fp8 MFMA QKV/conv"; the
 * reference has no fp8 path: torch autocast fp16 at most, scripts/stable_txt2img.py:711).  af_set_fp8(h, 1) on a bf16
 * handle: GroupNorm + SiLU (openaimodel.py:259-263, in_layers / out_layers) writes e4m3 and the 3x3 convolutions that read
 * it multiply on v_mfma_scale_f32_16x16x128_f8f6f4 with power-of-two scales (per output channel for the weights, 2^3 for
 * the activations); likewise norm1 -> to_q / to_k / to_v of every BasicTransformerBlock's self-attention
 * (attention.py:195-196, 275-285); everything else stays bf16.  Tolerance: tests/test_fp8_gpu.py. */
int af_set_fp8(af_handle* h, int on);
int64_t af_fp8_gemm_launches(void); /* launches on the fp8 kernel since af_gemm_plan_counts_reset */
int64_t af_halo8_launches(void);    /* launches of the eight-wave LDS-halo 3x3 kernel (also counted under tile 5) */
int64_t af_gn_producer_launches(void); /* convolutions that also wrote the GroupNorm partial sums of their output (no statistics pass in the consumer) */
/* launches of the register-resident short-key cross-attention kernel (bf16, <= 96 keys, dh 40 / 80) since the last
 * af_gemm_plan_counts_reset */
int64_t af_attn_short_launches(void);
/* launches of the one-kernel cross-attention layer (bf16, C = 320, 8 heads x 40, <= 80 keys: LayerNorm-folded to_q + attention +
 * to_out + residual; adaface_amd/csrc/af_xattn_fused.hip) since the last af_gemm_plan_counts_reset */
int64_t af_xattn_fused_launches(void);
/* the same layer as an operator (parity tests; /root/reference/ldm/modules/attention.py:172-257, 279): x [B, N, 320] fp32,
 * ln_stats [B * N][2] = (mean, rstd) of the bf16-rounded rows, gamma / beta [320], wq [320, 320] (to_q, no bias), kv [B, S, 640]
 * = the context's K | V projections, wo [320, 320] + bo [320] (to_out); y = x + to_out(softmax(to_q(LN(x)) K^T / sqrt(40)) V) */
int af_op_xattn_fused(const float* x_dev, const float* ln_stats_dev, const float* gamma_dev, const float* beta_dev,
                      const float* wq_dev, const float* kv_dev, const float* wo_dev, const float* bo_dev, float* y_dev,
                      float* ln_parts_out_dev, int B, int N, int S, void* stream);
/* row-panel GEMM launches that applied the GroupNorm of their input in their prologue (SpatialTransformer.norm + proj_in) */
int64_t af_gn_consumer_launches(void);
int64_t af_up_phase4_launches(void); /* upsampled 3x3 convolutions run as four 2x2 phase convolutions on the stored map */
int64_t af_rowpanel_launches(void); /* launches of the row-panel kernels (K = 320 / 640 / 1280 GEMMs with the activation rows resident in registers) */
int af_op_conv2d_fp8(const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev, float* y_dev,
                     int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int upsample, int act_shift,
                     void* stream);
int af_op_groupnorm_fp8(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, int silu,
                        unsigned char* y8_dev, int B, int C, int H, int W, int act_shift, void* stream);
int af_op_layernorm_fp8(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, unsigned char* y8_dev,
                        int64_t rows, int C, int act_shift, void* stream);
/* GroupNorm(32) (no SiLU) + 1x1 convolution (SpatialTransformer.norm + proj_in, attention.py:325-326) on the same bf16
 * operands both ways: y_plain = apply pass + GEMM, y_fused = row-panel GEMM that normalises its rows in its prologue.
 * x [B,C,H,W], w [N,C], outputs [B,N,H,W] fp32; fails when the shape has no row-panel launch. */
int af_op_gn_conv1x1(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, const float* w_dev,
                     const float* bias_dev, float* y_plain_dev, float* y_fused_dev, int B, int C, int H, int W, int N,
                     void* stream);
/* F.layer_norm over the last dim of [rows, C]. */
int af_op_layernorm(int dtype, const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps,
                    float* y_dev, int64_t rows, int C, void* stream);
/* multi-head attention on [B,N,heads*dh] / [B,S,heads*dh] tensors (attention.py:197-243); `causal` is a flag word:
 * bit 0: query i sees keys <= i (CLIP text tower); bit 1 (test hook): K and V are placed at the head of allocations
 * whose 128 tail rows hold NaNs, so reads past row S of the last sample become visible in the output. */
int af_op_attention(int dtype, const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int B, int Nq,
                    int Nk, int heads, int dh, float scale, int causal, void* stream);
/* timestep_embedding (util.py:154-174): t [B] int64 -> y [B,dim] fp32. */
int af_op_timestep_embedding(int dtype, const int64_t* t_dev, float* y_dev, int B, int dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif
