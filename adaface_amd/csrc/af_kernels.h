// Internal launcher declarations (one template per storage type T = bf16 | float).
#pragma once
#include "af_common.h"

AfGemmPlan af_plan_conv_gemm(const ConvGemmParams& p, int batch, int elem_size);
template <typename T>
int af_launch_conv_gemm(const ConvGemmParams& p, int batch, hipStream_t stream, const AfGemmPlan* plan = nullptr,
                        void* ws = nullptr);
template <typename T> int af_launch_attention(const AttnParams& p, int B, int dh, hipStream_t stream);
// V packed as the resident MFMA fragments of the short-key cross-attention kernel (bf16; Nk <= 96, dh 40 / 80): elements
// needed (0 = no such kernel for this case) and the pack launch; AttnParams::vt_pack then selects the kernel
template <typename T> long af_attn_short_pack_elems(int B, int H, int dh, int Nk);
template <typename T>
int af_launch_attn_short_pack(const void* v, int ldv, long bsv, int Nk, int H, int dh, int B, void* vt, hipStream_t stream);
extern std::atomic<long> g_af_attn_short_launches;
// ONE launch per cross-attention layer (bf16, C = 320, 8 heads x 40, <= 80 keys; af_xattn_fused.hip): LayerNorm-folded to_q,
// attention over the packed context K / V, to_out + bias + residual, LayerNorm partial sums for the next consumer
struct AfXattnFusedParams {
  const void* x; int ldx; int M; int rows_per_sample;
  const float* ln_stats; int ln_parts_n; float ln_inv_count, ln_eps;   // as ConvGemmParams::ln_stats / ln_parts_n
  const void* wq; int ldwq; const float* q_colsum; const float* q_bias;
  const void* kvpack;                                                  // af_launch_xattn_fused_pack
  const void* wo; int ldwo; const float* o_bias;                       // K-permuted to_out weight (af_launch_xattn_fused_permute_wo)
  void* out; int ldo;
  float* ln_stats_out;                                                 // [4][M][2] or null
  int Nk;
};
bool af_xattn_fused_ok(int M, int rows_per_sample, int C, int H, int dh, int Nk);
long af_xattn_fused_pack_elems(int B, int H, int dh, int Nk);         // bf16 elements of the K / V pack of one layer (0 = no such kernel)
int af_launch_xattn_fused_pack(const void* kv, int ldk, long bsk, int Nk, int B, void* pack, hipStream_t stream);   // K as stored: the kernel applies dh^-1/2 * log2 e in fp32
int af_launch_xattn_fused_permute_wo(const void* w, int ldw, int rows, void* wp, int ldp, hipStream_t stream);
int af_launch_xattn_fused(const AfXattnFusedParams& a, hipStream_t stream);
extern std::atomic<long> g_af_xattn_fused_launches;

size_t af_gn_workspace_bytes(int B, int HW);
template <typename T>
int af_launch_groupnorm(const void* x, long x_bs, int ldx, int B, int HW, int Cn, const float* gamma,
                        const float* beta, float eps, int silu, void* y, long y_bs, int ldy, void* workspace,
                        hipStream_t stream, float fp8_mul = 0.f,    // fp8_mul != 0: y is e4m3 bytes of result * fp8_mul
                        const float* pre_partial = nullptr, int pre_npart = 0);   // statistics already summed by the producer
// GroupNorm reduced to its per-sample affine map ab_out [B][2][Cn] (scale, shift) for a consumer that applies it itself
// (ConvGemmParams::gn_ab): statistics pass (unless pre_partial) + fold, no pass that writes the normalised tensor
template <typename T>
int af_launch_groupnorm_fold(const void* x, long x_bs, int ldx, int B, int HW, int Cn, const float* gamma, const float* beta,
                             float eps, void* workspace, hipStream_t stream, const float* pre_partial, int pre_npart,
                             float* ab_out);
// would a bf16 convolution with these parameters on this plan write GroupNorm partial sums (ConvGemmParams::gn_stats_out)?
bool af_conv_gn_stats_ok(const ConvGemmParams& p, const AfGemmPlan& pl, int cpg);
// the 8 x 8-map 3x3 convolution kernel (af_conv_s8.hip): four whole images x 80 columns per tile, always four K slices
bool af_conv_s8_ok(const ConvGemmParams& p, int batch);
int af_conv_s8_slices(const ConvGemmParams& p, int batch);    // K slices it runs in (8 x 8 maps: 4, 16 x 16 maps: 1), 0 = not taken
int af_launch_conv_s8(const ConvGemmParams& p, hipStream_t stream);
int af_conv_rowpanel_kind(const ConvGemmParams& p, int batch);   // 0 = not a row-panel launch (p.splitk as planned)
template <typename T>
int af_launch_layernorm(const void* x, int ldx, long rows, int Cn, const float* gamma, const float* beta,
                        float eps, void* y, int ldy, hipStream_t stream, float fp8_mul = 0.f);

template <typename T>
int af_launch_nchw_to_nhwc(const float* x, void* y, int B, int Cn, int HW, int Cpad, float scale, hipStream_t s);
template <typename T> int af_launch_nhwc_to_nchw(const void* x, float* y, int B, int Cn, int HW, int ld, hipStream_t s);
template <typename T> int af_launch_cast_f32(const float* x, void* y, long n, hipStream_t s);
// fp32 [rows][D] -> T with dst row r taken from src row rowmap[r]
template <typename T>
int af_launch_gather_rows_cast(const float* x, const int* rowmap, void* y, long rows, int D, hipStream_t s);
// subject-token ks x ks conv attention (ldm/util.py:701-879; ks = 2, 3, 4) for B samples whose ks^2 subject keys are rows
// tok0..tok0+ks^2-1 of kv [S][2*H*dh] (K | V): o holds the flash result over the keys in front of the subject tokens (its
// log2-domain log-sum-exp in lse [B][H][N]) and is overwritten by the merged attention output, lse by the merged
// log-sum-exp (several subject strings per sample merge one after the other); sN = scratch [B][H][N][ks^2] floats
template <typename T>
int af_launch_conv_attn(const void* q, int ldq, long bsq, const void* kv, int ldk, long bsk, int tok0, float* sN,
                        float* lse, void* o, int ldo, long bso, int B, int N, int H, int dh, int Hh, int Ww,
                        float scale, int ks, hipStream_t s);
template <typename T> int af_launch_cast_to_f32(const void* x, float* y, long n, hipStream_t s);
template <typename T> int af_launch_timestep_embedding(const long long* t, void* y, int B, int dim, hipStream_t s);
template <typename T> int af_launch_silu(const void* x, void* y, long n, hipStream_t s);
// CLIP text tower pieces (encoders/modules.py:179-463; transformers CLIPTextModel)
template <typename T> int af_launch_quick_gelu(const void* x, void* y, long n, hipStream_t s);
template <typename T>
int af_launch_embed_rows(const long long* ids, const void* table, int vocab, int D, float* y, long n, hipStream_t s);
template <typename T>
int af_launch_add_pos_cast(const float* x, const void* pos, int Tn, int D, void* y, long rows, hipStream_t s);
template <typename T> int af_launch_blend2(const void* a, float w0, const void* b, float w1, void* y, long n, hipStream_t s);
template <typename T>
int af_launch_copy_channels(const void* src, int lds_, void* dst, int ldd, int off, int Cn, long npix, hipStream_t s);
int af_launch_ddim_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, long n,
                        float guidance, float a_t, float a_prev, float sqrt_one_minus_at, float sigma_t,
                        float temperature, float* x_prev, float* pred_x0, hipStream_t s);
int af_launch_posterior_sample(const float* mom, const float* noise, float scale, float* z, int B, int Cn, long HW,
                               hipStream_t s);
int af_launch_lincomb(float* out, long n, const float* x0, float w0, const float* x1, float w1, const float* x2, float w2,
                      const float* x3, float w3, int mode, hipStream_t s);
template <typename T> int af_launch_softmax_rows(void* x, int ld, int ncols, long rows, hipStream_t s);
template <typename T>
int af_launch_transpose(const void* x, long x_bs, int ldx, void* y, long y_bs, int R, int Cn, int B, hipStream_t s);
template <typename T> int af_launch_to_uint8(const void* x, int ld, uint8_t* y, long npix, hipStream_t s);
int af_launch_nchw_to_uint8(const float* x, uint8_t* y, int B, int HW, hipStream_t s);

// weight repack (device): src fp32 [rows][cin][ks][ks] -> dst T rows [row_off + perm(n)][ks*ks*cin_pad]
//   perm: 0 identity, 1 GEGLU interleave (value/gate groups of 16, half = rows/2)
template <typename T>
int af_launch_repack_weight(const float* src, void* dst, int rows, int cin, int cin_pad, int ks, int ldw, int row_off,
                            int perm, hipStream_t s);
int af_launch_permute_bias(const float* src, float* dst, int rows, int perm, hipStream_t s);
// LayerNorm folded into a linear: Wf = W * gamma (rounded to T), colsum[n] = sum_k Wf[n][k], biasf = W beta + bias
template <typename T>
int af_launch_ln_fold(const void* W, void* Wf, const float* gamma, const float* beta, const float* bias, float* colsum,
                      float* biasf, int rows, int K, int ldw, hipStream_t s);

// LayerNorm row statistics [parts][M][2] (sum, sum of squares) -> [M][2] (mu, rstd)
int af_launch_ln_finalize(const float* part, int parts, int M, int count, float eps, float* out, hipStream_t s);

// plan of the most recent af_launch_conv_gemm (diagnostics, af_last_gemm_plan)
void af_set_last_plan(const AfGemmPlan& pl);
AfGemmPlan af_get_last_plan();
extern std::atomic<long> g_af_plan_counts[15];
extern std::atomic<long> g_af_gn_consumer_launches;
int af_launch_up_phase4_weights(const void* w3, int rows, int cin, int ldw3, void* w4, hipStream_t stream);
// fp8 (e4m3) twin of a repacked bf16 weight (K in 64-channel units, power-of-two row scales) / saturating bf16 -> e4m3 cast
int af_launch_quant_weight_fp8(const void* w, int rows, int ldw, int cin_pad, int ks, void* w8, int k8, unsigned char* sc,
                               hipStream_t stream);
int af_launch_cast_fp8(const void* x, void* y, long n, float mul, hipStream_t stream);
