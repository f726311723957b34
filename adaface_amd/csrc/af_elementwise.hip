// Small HBM-bound kernels around the contractions: layout conversion at the C-ABI
// boundary (the reference speaks NCHW fp32, the kernels NHWC T), timestep embedding
// (ldm/modules/diffusionmodules/util.py:154-174), the fused CFG + DDIM update
// (ldm/models/diffusion/ddim.py:260,273-295), row softmax for the VAE's single-head
// attention (ldm/modules/diffusionmodules/model.py:222-236), channel concat
// (openaimodel.py:1019) and the image post-process (scripts/stable_txt2img.py:715,764-765).
#include "af_common.h"
#include <math.h>

#define EW_GRID(n) dim3((unsigned)(((n) + 255) / 256 > 65535 * 16 ? 65535 * 16 : ((n) + 255) / 256))

// NCHW fp32 [B,C,H,W] -> NHWC T [B,H*W,Cpad] (channels >= C zero), scaled
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int Cn, int HW,
                                    int Cpad, float scale) {
  const long n = (long)B * HW * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const long bp = i / Cpad;
    const int pix = (int)(bp % HW);
    const long b = bp / HW;
    float v = 0.f;
    if (c < Cn) v = x[(b * Cn + c) * HW + pix] * scale;
    y[i] = from_f32<T>(v);
  }
}

// NHWC T [B,HW,ld] (first C channels) -> NCHW fp32 [B,C,H,W]
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int Cn, int HW,
                                    int ld) {
  const long n = (long)B * Cn * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW);
    const long bc = i / HW;
    const int c = (int)(bc % Cn);
    const long b = bc / Cn;
    y[i] = to_f32<T>(x[(b * HW + pix) * ld + c]);
  }
}

// fp32 -> T cast, contiguous
template <typename T>
__global__ void cast_f32_kernel(const float* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(x[i]);
}
// fp32 -> T cast of [rows][D] with a per-row source map (set_context: subject tokens moved to the end of the key list)
template <typename T>
__global__ void gather_rows_cast_kernel(const float* __restrict__ x, const int* __restrict__ rowmap,
                                        T* __restrict__ y, long rows, int D) {
  const long n = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    y[i] = from_f32<T>(x[(long)rowmap[r] * D + (i - r * D)]);
  }
}

// ---- subject-token convolutional attention (ldm/util.py:701-879 replace_rows_by_conv_attn; kernel sizes 2, 3, 4) ----
// KS x KS taps, NT = KS^2 subject tokens in tap order (row-major).  The reference pads q by (left, right, top, bottom) =
// (0,1,0,1) / (1,1,1,1) / (1,2,1,2) for KS = 2 / 3 / 4 (util.py:747-760): tap (ty, tx) reads pixel (y + ty - P0, x + tx - P0)
// with P0 = the left / top pad, and subject token j = (jy, jx) receives the conv map shifted by (dy, dx) = (jy - P0, jx - P0)
// with zero fill (util.py:812-836): column_j(y, x) = A(y - dy, x - dx).
// Step 1: pointwise scores of the NT subject keys: sN[bb][h][p][t] = scale * q[b][p][h] . k[b][tok0+t][h]
// (the subject tokens sit at the END of the cached key list: set_context permutes them there; tok0 = their first row).
template <typename T, int KS>
__global__ __launch_bounds__(256) void subj_scores_kernel(const T* __restrict__ q, int ldq, long bsq,
                                                          const T* __restrict__ kv, int ldk, long bsk, int tok0,
                                                          float* __restrict__ sN, int N, int H, int dh, float scale) {
  constexpr int NT = KS * KS;
  __shared__ float ks[NT * 160];
  const int h = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < NT * dh; i += 256)
    ks[i] = to_f32<T>(kv[(long)b * bsk + (long)(tok0 + i / dh) * ldk + h * dh + (i % dh)]);
  __syncthreads();
  const int pp = blockIdx.x * 256 + threadIdx.x;
  if (pp >= N) return;
  const T* qp = q + (long)b * bsq + (long)pp * ldq + h * dh;
  float acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0.f;
  for (int c = 0; c < dh; c += 4) {
    Quad<T> qv;
    qv.load(qp + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = to_f32<T>(qv.e[e]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = fmaf(f, ks[t * dh + c + e], acc[t]);
    }
  }
  float* o = sN + (((long)b * H + h) * N + pp) * NT;
#pragma unroll
  for (int t = 0; t < NT; ++t) o[t] = acc[t] * scale;
}

// Step 2: conv scores A(y,x) = KS^-1.5 * sum_t sN[(y+ty-P0, x+tx-P0)][t] (zero outside the map), column j of the subject
// = A shifted by (dy,dx) = (j/KS-P0, j%KS-P0) with zero fill, then the exact softmax merge of the NT replaced keys with
// the flash result over the other S-NT keys:  out = (w_U O_U + sum_j e_j v_j) / (w_U + sum_j e_j),
// w_U = 2^(lse_U - M), e_j = 2^(r_j log2e - M).  o holds O_U on entry and the merged output on exit, lse likewise (a sample
// that carries several subject strings -- attention.py:208-216 loops over them -- merges them one after the other).
template <typename T, int KS>
__global__ __launch_bounds__(256) void conv_attn_merge_kernel(const float* __restrict__ sN, float* __restrict__ lse,
                                                              const T* __restrict__ kv, int ldk, long bsk, int tok0,
                                                              T* __restrict__ o, int ldo, long bso, int N, int H, int dh,
                                                              int Hh, int Ww) {
  constexpr int NT = KS * KS, P0 = KS == 2 ? 0 : 1;
  __shared__ float vs[NT * 160];
  const int h = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < NT * dh; i += 256)
    vs[i] = to_f32<T>(kv[(long)b * bsk + (long)(tok0 + i / dh) * ldk + H * dh + h * dh + (i % dh)]);   // V half
  __syncthreads();
  const int pp = blockIdx.x * 256 + threadIdx.x;
  if (pp >= N) return;
  const int y = pp / Ww, x = pp - y * Ww;
  const float* sb = sN + ((long)b * H + h) * N * NT;
  const float inv_norm = KS == 2 ? 0.35355339059327373f : (KS == 3 ? 0.19245008972987526f : 0.125f);   // KS^-1.5
  float r[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int yy = y - (j / KS - P0), xx = x - (j % KS - P0);
    float a = 0.f;
    if ((unsigned)yy < (unsigned)Hh && (unsigned)xx < (unsigned)Ww) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int y2 = yy + t / KS - P0, x2 = xx + t % KS - P0;
        if ((unsigned)y2 < (unsigned)Hh && (unsigned)x2 < (unsigned)Ww) a += sb[(long)(y2 * Ww + x2) * NT + t];
      }
    }
    r[j] = a * inv_norm * 1.44269504088896340736f;   // to the log2 domain of the flash kernel
  }
  const float L = lse[((long)b * H + h) * N + pp];
  float M = L;
#pragma unroll
  for (int j = 0; j < NT; ++j) M = fmaxf(M, r[j]);
  const float wU = exp2f(L - M);
  float den = wU;
#pragma unroll
  for (int j = 0; j < NT; ++j) { r[j] = exp2f(r[j] - M); den += r[j]; }
  const float inv = 1.0f / den;
  lse[((long)b * H + h) * N + pp] = M + log2f(den);   // log-sum-exp including these NT keys: the next subject string merges onto it
  T* op = o + (long)b * bso + (long)pp * ldo + h * dh;
  for (int c = 0; c < dh; c += 4) {
    Quad<T> ov;
    ov.load(op + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = wU * to_f32<T>(ov.e[e]);
#pragma unroll
      for (int j = 0; j < NT; ++j) v = fmaf(r[j], vs[j * dh + c + e], v);
      ov.e[e] = from_f32<T>(v * inv);
    }
    ov.store(op + c);
  }
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = to_f32<T>(x[i]);
}

// timestep_embedding: emb[b] = [cos(t*f_i) | sin(t*f_i)], f_i = exp(-ln(max_period)*i/half)
// (util.py:163-169; cosine half FIRST).  Output T [B, dim].
template <typename T>
__global__ void timestep_embedding_kernel(const long long* __restrict__ t, T* __restrict__ y, int B, int dim) {
  const int half = dim / 2;
  const int n = B * dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i % dim;
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      const float freq = expf(-9.210340371976184f * (float)k / (float)half);
      const float arg = (float)t[b] * freq;
      v = j < half ? cosf(arg) : sinf(arg);
    }
    y[i] = from_f32<T>(v);
  }
}

// y = silu(x) elementwise (T)
template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float f = to_f32<T>(x[i]);
    y[i] = from_f32<T>(f / (1.0f + expf(-f)));
  }
}

// y = x * sigmoid(1.702 x) ("quick_gelu", the CLIP text tower's activation: transformers CLIPMLP, hidden_act of
// openai/clip-vit-large-patch14) elementwise (T)
template <typename T>
__global__ void quick_gelu_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float f = to_f32<T>(x[i]);
    y[i] = from_f32<T>(f / (1.0f + expf(-1.702f * f)));
  }
}
// CLIPTextEmbeddings (as patched by encoders/modules.py:198-227): rows of the token table -> fp32 [n, D] (the
// EmbeddingManager patches these on the caller's side), then  x = T(inputs_embeds + position_embedding[pos])
template <typename T>
__global__ void embed_rows_kernel(const long long* __restrict__ ids, const T* __restrict__ table, int vocab, int D,
                                  float* __restrict__ y, long n) {
  const long total = n * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    long long id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    y[i] = to_f32<T>(table[id * D + c]);
  }
}
template <typename T>
__global__ void add_pos_cast_kernel(const float* __restrict__ x, const T* __restrict__ pos, int Tn, int D,
                                    T* __restrict__ y, long rows) {
  const long total = rows * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    y[i] = from_f32<T>(x[i] + to_f32<T>(pos[(r % Tn) * D + c]));
  }
}
// y (fp32) = w0 * a + w1 * b   (last-layers blend of the CLIP hidden states, encoders/modules.py:361-368), T inputs
template <typename T>
__global__ void blend2_kernel(const T* __restrict__ a, float w0, const T* __restrict__ b, float w1, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = from_f32<T>(w0 * to_f32<T>(a[i]) + w1 * to_f32<T>(b[i]));
}

// dst[pix][off + c] = src[pix][c]   (channel concat; 16-byte vectors)
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, int off,
                                     int Cn, long npix) {
  constexpr int EPC = 16 / sizeof(T);
  const int NV = Cn / EPC;
  const long n = npix * NV;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / NV;
    const int v = (int)(i % NV);
    *reinterpret_cast<uint4*>(dst + pix * ldd + off + v * EPC) =
        *reinterpret_cast<const uint4*>(src + pix * lds_ + v * EPC);
  }
}

// Fused classifier-free guidance + DDIM update, fp32 NCHW, n = B*C*H*W elements.
//   e = e_u + g (e_c - e_u);  pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t)
//   x_prev = sqrt(a_prev) pred_x0 + sqrt(1-a_prev-sigma^2) e + sigma*temperature*noise
// eps holds [cond ; uncond] halves (ddim.py:243,252: cond FIRST).
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                 const float* __restrict__ eps_u, const float* __restrict__ noise, long n,
                                 float guidance, float a_t, float a_prev, float sqrt_one_minus_at, float sigma_t,
                                 float temperature, float* __restrict__ x_prev, float* __restrict__ pred_x0) {
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_c = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float ec = eps_c[i];
    float e = ec;
    if (eps_u) {
      const float eu = eps_u[i];
      e = eu + guidance * (ec - eu);
    }
    const float xv = x[i];
    const float p0 = (xv - sqrt_one_minus_at * e) / sqrt_at;
    float xp = sqrt_aprev * p0 + dir_c * e;
    if (noise) xp += sigma_t * noise[i] * temperature;
    x_prev[i] = xp;
    if (pred_x0) pred_x0[i] = p0;
  }
}

// out = w0*x0 + w1*x1 + w2*x2 + w3*x3 (unused inputs NULL), fp32.  mode 1: out = x1 + w0*(x0 - x1), the classifier-
// free-guidance form e_u + g (e_c - e_u) (ddim.py:260, plms.py:199).  Serves PLMS's Adams-Bashforth combinations
// of the last noise predictions (plms.py:236-249).
__global__ void lincomb_kernel(float* __restrict__ out, long n, const float* __restrict__ x0, float w0,
                               const float* __restrict__ x1, float w1, const float* __restrict__ x2, float w2,
                               const float* __restrict__ x3, float w3, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v;
    if (mode == 1) {
      const float a = x1[i];
      v = a + w0 * (x0[i] - a);
    } else {
      v = w0 * x0[i];
      if (x1) v += w1 * x1[i];
      if (x2) v += w2 * x2[i];
      if (x3) v += w3 * x3[i];
    }
    out[i] = v;
  }
}

// DiagonalGaussianDistribution.sample (distributions.py:27-37) + the scale_factor of get_first_stage_encoding
__global__ void posterior_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float scale,
                                        float* __restrict__ z, int Cn, long HW, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (Cn * HW), r = i - b * (Cn * HW);
    const float mean = mom[b * 2 * Cn * HW + r];
    float v = mean;
    if (noise) {
      const float lv = fminf(fmaxf(mom[b * 2 * Cn * HW + Cn * HW + r], -30.0f), 20.0f);
      v = mean + expf(0.5f * lv) * noise[i];
    }
    z[i] = scale * v;
  }
}

// row softmax in place over fp32-accumulated T rows: x[row][0..n) (VAE AttnBlock)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ x, int ld, int ncols, long rows) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  if (row >= rows) return;
  T* xr = x + row * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int c = tid; c < ncols; c += 256) mx = fmaxf(mx, to_f32<T>(xr[c]));
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int c = tid; c < ncols; c += 256) s += expf(to_f32<T>(xr[c]) - mx);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / s;
  for (int c = tid; c < ncols; c += 256) xr[c] = from_f32<T>(expf(to_f32<T>(xr[c]) - mx) * inv);
}

// y[b][c][r] = x[b][r][c]   (x: [B][R][ldx] take first Cn columns; y: [B][Cn][R])
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ x, long x_bs, int ldx, T* __restrict__ y, long y_bs,
                                 int R, int Cn) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cn) ? to_f32<T>(x[(long)b * x_bs + (long)r * ldx + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < Cn) y[(long)b * y_bs + (long)c * R + r] = from_f32<T>(tile[tx][i]);
  }
}

// image post-process: NHWC T [B,HW,ld] (3 channels) -> uint8 HWC [B,HW,3]:
//   clamp((x+1)/2, 0, 1) * 255, truncated (numpy astype(uint8)), stable_txt2img.py:715,764-765
template <typename T>
__global__ void to_uint8_kernel(const T* __restrict__ x, int ld, uint8_t* __restrict__ y, long npix) {
  const long n = npix * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / 3;
    const int c = (int)(i % 3);
    float f = (to_f32<T>(x[pix * ld + c]) + 1.0f) * 0.5f;
    f = fminf(fmaxf(f, 0.f), 1.f);
    y[i] = (uint8_t)(255.f * f);
  }
}
// same post-process from an fp32 NCHW image [B,3,H,W] -> uint8 HWC
__global__ void nchw_to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int B, int HW) {
  const long n = (long)B * HW * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const long bp = i / 3;
    const int pix = (int)(bp % HW);
    const long b = bp / HW;
    float f = (x[(b * 3 + c) * HW + pix] + 1.0f) * 0.5f;
    f = fminf(fmaxf(f, 0.f), 1.f);
    y[i] = (uint8_t)(255.f * f);
  }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
template <typename T>
int af_launch_nchw_to_nhwc(const float* x, void* y, int B, int Cn, int HW, int Cpad, float scale, hipStream_t s) {
  const long n = (long)B * HW * Cpad;
  hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), EW_GRID(n), dim3(256), 0, s, x, reinterpret_cast<T*>(y), B, Cn, HW,
                     Cpad, scale);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_nhwc_to_nchw(const void* x, float* y, int B, int Cn, int HW, int ld, hipStream_t s) {
  const long n = (long)B * Cn * HW;
  hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(x), y, B, Cn,
                     HW, ld);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_cast_f32(const float* x, void* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((cast_f32_kernel<T>), EW_GRID(n), dim3(256), 0, s, x, reinterpret_cast<T*>(y), n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_gather_rows_cast(const float* x, const int* rowmap, void* y, long rows, int D, hipStream_t s) {
  hipLaunchKernelGGL((gather_rows_cast_kernel<T>), EW_GRID(rows * D), dim3(256), 0, s, x, rowmap, reinterpret_cast<T*>(y),
                     rows, D);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T, int KS>
static int launch_conv_attn_ks(const void* q, int ldq, long bsq, const void* kv, int ldk, long bsk, int tok0, float* sN,
                               float* lse, void* o, int ldo, long bso, int B, int N, int H, int dh, int Hh, int Ww,
                               float scale, hipStream_t s) {
  dim3 grid((N + 255) / 256, H, B);
  hipLaunchKernelGGL((subj_scores_kernel<T, KS>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(q), ldq, bsq,
                     reinterpret_cast<const T*>(kv), ldk, bsk, tok0, sN, N, H, dh, scale);
  hipLaunchKernelGGL((conv_attn_merge_kernel<T, KS>), grid, dim3(256), 0, s, sN, lse, reinterpret_cast<const T*>(kv), ldk, bsk,
                     tok0, reinterpret_cast<T*>(o), ldo, bso, N, H, dh, Hh, Ww);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_conv_attn(const void* q, int ldq, long bsq, const void* kv, int ldk, long bsk, int tok0, float* sN,
                        float* lse, void* o, int ldo, long bso, int B, int N, int H, int dh, int Hh, int Ww,
                        float scale, int ks, hipStream_t s) {
  if (dh > 160 || dh % 4 != 0 || Hh * Ww != N || tok0 <= 0) return -1;
  switch (ks) {
    case 2: return launch_conv_attn_ks<T, 2>(q, ldq, bsq, kv, ldk, bsk, tok0, sN, lse, o, ldo, bso, B, N, H, dh, Hh, Ww, scale, s);
    case 3: return launch_conv_attn_ks<T, 3>(q, ldq, bsq, kv, ldk, bsk, tok0, sN, lse, o, ldo, bso, B, N, H, dh, Hh, Ww, scale, s);
    case 4: return launch_conv_attn_ks<T, 4>(q, ldq, bsq, kv, ldk, bsk, tok0, sN, lse, o, ldo, bso, B, N, H, dh, Hh, Ww, scale, s);
    default: return -1;
  }
}
template <typename T> int af_launch_cast_to_f32(const void* x, float* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((cast_to_f32_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(x), y, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_timestep_embedding(const long long* t, void* y, int B, int dim, hipStream_t s) {
  hipLaunchKernelGGL((timestep_embedding_kernel<T>), EW_GRID((long)B * dim), dim3(256), 0, s, t,
                     reinterpret_cast<T*>(y), B, dim);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_silu(const void* x, void* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((silu_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(x),
                     reinterpret_cast<T*>(y), n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_quick_gelu(const void* x, void* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((quick_gelu_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(x),
                     reinterpret_cast<T*>(y), n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_embed_rows(const long long* ids, const void* table, int vocab, int D, float* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((embed_rows_kernel<T>), EW_GRID(n * D), dim3(256), 0, s, ids, reinterpret_cast<const T*>(table), vocab,
                     D, y, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_add_pos_cast(const float* x, const void* pos, int Tn, int D, void* y, long rows, hipStream_t s) {
  hipLaunchKernelGGL((add_pos_cast_kernel<T>), EW_GRID(rows * D), dim3(256), 0, s, x, reinterpret_cast<const T*>(pos), Tn, D,
                     reinterpret_cast<T*>(y), rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_blend2(const void* a, float w0, const void* b, float w1, void* y, long n, hipStream_t s) {
  hipLaunchKernelGGL((blend2_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(a), w0,
                     reinterpret_cast<const T*>(b), w1, reinterpret_cast<T*>(y), n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_copy_channels(const void* src, int lds_, void* dst, int ldd, int off, int Cn, long npix, hipStream_t s) {
  constexpr int EPC = 16 / sizeof(T);
  if (Cn % EPC || lds_ % EPC || ldd % EPC || off % EPC) {
    af_set_error_msg("copy_channels: misaligned C=%d", Cn);
    return -1;
  }
  const long n = npix * (Cn / EPC);
  hipLaunchKernelGGL((copy_channels_kernel<T>), EW_GRID(n), dim3(256), 0, s, reinterpret_cast<const T*>(src), lds_,
                     reinterpret_cast<T*>(dst), ldd, off, Cn, npix);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int af_launch_ddim_step(const float* x, const float* eps_c, const float* eps_u, const float* noise, long n,
                        float guidance, float a_t, float a_prev, float sqrt_one_minus_at, float sigma_t,
                        float temperature, float* x_prev, float* pred_x0, hipStream_t s) {
  hipLaunchKernelGGL(ddim_step_kernel, EW_GRID(n), dim3(256), 0, s, x, eps_c, eps_u, noise, n, guidance, a_t, a_prev,
                     sqrt_one_minus_at, sigma_t, temperature, x_prev, pred_x0);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int af_launch_lincomb(float* out, long n, const float* x0, float w0, const float* x1, float w1, const float* x2, float w2,
                      const float* x3, float w3, int mode, hipStream_t s) {
  hipLaunchKernelGGL(lincomb_kernel, EW_GRID(n), dim3(256), 0, s, out, n, x0, w0, x1, w1, x2, w2, x3, w3, mode);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int af_launch_posterior_sample(const float* mom, const float* noise, float scale, float* z, int B, int Cn, long HW,
                               hipStream_t s) {
  const long total = (long)B * Cn * HW;
  hipLaunchKernelGGL(posterior_sample_kernel, EW_GRID(total), dim3(256), 0, s, mom, noise, scale, z, Cn, HW, total);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_softmax_rows(void* x, int ld, int ncols, long rows, hipStream_t s) {
  hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)rows), dim3(256), 0, s, reinterpret_cast<T*>(x), ld,
                     ncols, rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T>
int af_launch_transpose(const void* x, long x_bs, int ldx, void* y, long y_bs, int R, int Cn, int B, hipStream_t s) {
  dim3 grid((R + 31) / 32, (Cn + 31) / 32, B);
  hipLaunchKernelGGL((transpose_kernel<T>), grid, dim3(256), 0, s, reinterpret_cast<const T*>(x), x_bs, ldx,
                     reinterpret_cast<T*>(y), y_bs, R, Cn);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T> int af_launch_to_uint8(const void* x, int ld, uint8_t* y, long npix, hipStream_t s) {
  hipLaunchKernelGGL((to_uint8_kernel<T>), EW_GRID(npix * 3), dim3(256), 0, s, reinterpret_cast<const T*>(x), ld, y,
                     npix);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int af_launch_nchw_to_uint8(const float* x, uint8_t* y, int B, int HW, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_uint8_kernel, EW_GRID((long)B * HW * 3), dim3(256), 0, s, x, y, B, HW);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

#define INST(T)                                                                                               \
  template int af_launch_nchw_to_nhwc<T>(const float*, void*, int, int, int, int, float, hipStream_t);        \
  template int af_launch_nhwc_to_nchw<T>(const void*, float*, int, int, int, int, hipStream_t);               \
  template int af_launch_cast_f32<T>(const float*, void*, long, hipStream_t);                                 \
  template int af_launch_gather_rows_cast<T>(const float*, const int*, void*, long, int, hipStream_t);         \
  template int af_launch_conv_attn<T>(const void*, int, long, const void*, int, long, int, float*, float*, \
                                      void*, int, long, int, int, int, int, int, int, float, int, hipStream_t); \
  template int af_launch_cast_to_f32<T>(const void*, float*, long, hipStream_t);                              \
  template int af_launch_timestep_embedding<T>(const long long*, void*, int, int, hipStream_t);               \
  template int af_launch_silu<T>(const void*, void*, long, hipStream_t);                                      \
  template int af_launch_quick_gelu<T>(const void*, void*, long, hipStream_t);                                \
  template int af_launch_embed_rows<T>(const long long*, const void*, int, int, float*, long, hipStream_t);   \
  template int af_launch_add_pos_cast<T>(const float*, const void*, int, int, void*, long, hipStream_t);      \
  template int af_launch_blend2<T>(const void*, float, const void*, float, void*, long, hipStream_t);         \
  template int af_launch_copy_channels<T>(const void*, int, void*, int, int, int, long, hipStream_t);         \
  template int af_launch_softmax_rows<T>(void*, int, int, long, hipStream_t);                                 \
  template int af_launch_transpose<T>(const void*, long, int, void*, long, int, int, int, hipStream_t);       \
  template int af_launch_to_uint8<T>(const void*, int, uint8_t*, long, hipStream_t);
INST(bf16)
INST(float)

// ---------------------------------------------------------------------------
// weight repack: [rows][cin][ks][ks] fp32 -> T [row_off + perm(n)][(ky,kx,c) with c padded to cin_pad]
// ---------------------------------------------------------------------------
__device__ __forceinline__ int geglu_perm(int n, int half) {
  const int j = n < half ? n : n - half;
  return (j >> 4) * 32 + (n < half ? 0 : 16) + (j & 15);
}
template <typename T>
__global__ void repack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int rows, int cin,
                                     int cin_pad, int ks, int ldw, int row_off, int perm) {
  const int kk = ks * ks;
  const long n = (long)rows * kk * cin_pad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cin_pad);
    const long t = i / cin_pad;
    const int tap = (int)(t % kk);
    const int r = (int)(t / kk);
    float v = 0.f;
    if (c < cin) v = src[((long)r * cin + c) * kk + tap];
    const int rr = perm ? geglu_perm(r, rows >> 1) : r;
    dst[(long)(row_off + rr) * ldw + tap * cin_pad + c] = from_f32<T>(v);
  }
}
__global__ void permute_bias_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int perm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x)
    dst[perm ? geglu_perm(i, rows >> 1) : i] = src[i];
}
template <typename T>
int af_launch_repack_weight(const float* src, void* dst, int rows, int cin, int cin_pad, int ks, int ldw, int row_off,
                            int perm, hipStream_t s) {
  const long n = (long)rows * ks * ks * cin_pad;
  hipLaunchKernelGGL((repack_weight_kernel<T>), EW_GRID(n), dim3(256), 0, s, src, reinterpret_cast<T*>(dst), rows, cin,
                     cin_pad, ks, ldw, row_off, perm);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// LayerNorm folded into the following linear (ping-pong GEMM, LNMODE 1):  LN(x) W^T + b
//   = rstd * (x (W gamma)^T - mu * colsum) + (W beta + b)   with colsum[n] = sum_k (W gamma)[n][k].
// One wave per weight row: Wf[n][k] = T(W[n][k] * gamma[k]); colsum from the ROUNDED products (what the MFMA multiplies).
template <typename T>
__global__ void ln_fold_kernel(const T* __restrict__ W, T* __restrict__ Wf, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ bias, float* __restrict__ colsum,
                               float* __restrict__ biasf, int rows, int K, int ldw) {
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= rows) return;
  float cs = 0.f, bs = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = to_f32<T>(W[(long)n * ldw + k]);
    const T wf = from_f32<T>(w * gamma[k]);
    Wf[(long)n * ldw + k] = wf;
    cs += to_f32<T>(wf);
    bs += w * beta[k];
  }
  for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o, 64); bs += __shfl_xor(bs, o, 64); }
  if (lane == 0) { colsum[n] = cs; biasf[n] = bs + (bias ? bias[n] : 0.f); }
}
template <typename T>
int af_launch_ln_fold(const void* W, void* Wf, const float* gamma, const float* beta, const float* bias, float* colsum,
                      float* biasf, int rows, int K, int ldw, hipStream_t s) {
  hipLaunchKernelGGL((ln_fold_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, reinterpret_cast<const T*>(W),
                     reinterpret_cast<T*>(Wf), gamma, beta, bias, colsum, biasf, rows, K, ldw);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template int af_launch_ln_fold<bf16>(const void*, void*, const float*, const float*, const float*, float*, float*, int, int, int, hipStream_t);
template int af_launch_ln_fold<float>(const void*, void*, const float*, const float*, const float*, float*, float*, int, int, int, hipStream_t);

int af_launch_permute_bias(const float* src, float* dst, int rows, int perm, hipStream_t s) {
  hipLaunchKernelGGL(permute_bias_kernel, EW_GRID(rows), dim3(256), 0, s, src, dst, rows, perm);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template int af_launch_repack_weight<bf16>(const float*, void*, int, int, int, int, int, int, int, hipStream_t);
template int af_launch_repack_weight<float>(const float*, void*, int, int, int, int, int, int, int, hipStream_t);
