// adaface_amd — 3x3 / stride-1 convolution on 8 x 8 maps (bf16): the UNet's deepest level, 1280 -> 1280 (x12 per forward) and
// 2560 -> 1280 (x3) at Bf = 16, i.e. M = 1024 rows, N = 1280, K = 11520 / 23040
// (/root/reference/ldm/modules/diffusionmodules/openaimodel.py:208,234 ResBlock in_layers / out_layers; middle block :620-650).
//
// Round 3 ran these on the gathering eight-wave kernel: 256 x 160 tiles over EIGHT K slices (32 tiles are all there is), every
// activation pixel staged nine times, 1.2 MB of LDS-DMA per workgroup, 42 MB of fp32 slabs + a reduce launch: 50 us for 23.6
// GFLOP.  Here a tile is FOUR WHOLE IMAGES (4 x 64 = 256 rows) x 80 columns: 4 x 16 = 64 tiles over FOUR K slices of whole
// 64-channel chunks = 256 workgroups, and the (8 + 2)^2-pixel halos of the four images -- 400 pixels, exactly the halo buffer of
// conv3x3_halo8_kernel -- are staged ONCE per chunk; the nine taps read their MFMA operand from it at shifted pixel addresses.
// Per workgroup 0.7 MB of LDS-DMA, half the slab traffic.
//   waves: 8 x (32 rows x 80 columns) = 2 x 5 MFMA 16x16x32 blocks, 20 MFMAs per wave and tap step
//   LDS: two halo buffers of 448 pixel slots x 128 B (chunk c / c + 1; 400 pixels + one dummy piece so that every wave stages
//        seven pieces per chunk) + four weight slots of 80 x 128 B (prefetch distance three steps)
//   K slice = Cin / 64 / 4 chunks; fp32 slabs ws[slice][M][N], reduced (+ bias, time-embedding row, residual) by the launcher's
//   splitk_reduce_kernel, slabs summed in slice order: deterministic.
#include "af_kernels.h"

#include <type_traits>

namespace cs8 {
typedef bf16 T;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int BM = 256, BN = 80, NI = 5, MI = 2, SPLITK = 4;
constexpr int HSLOTS = 448, HBUF = HSLOTS * 128, NHQ = HSLOTS / 8 / 8;     // 56 pieces of 8 pixels: 7 per wave and chunk
constexpr int WBYTES = BN * 128, WPIECES = BN / 8, D = 3, NS = 4;
constexpr int W_BASE = 2 * HBUF, LDS_BYTES = W_BASE + NS * WBYTES;         // 114688 + 40960 = 155648
static_assert(LDS_BYTES <= 160 * 1024 && NHQ == 7, "LDS plan");

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}
__device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read128o(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkm4(u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(512) void conv3x3_s8_kernel(const ConvGemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int ntn = p.N / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;   // (consecutive ids share the row tile = its halo: same XCD or not, L2 or MALL)
  const int m0 = tm * BM, n0 = tn * BN, zk = blockIdx.z;
  const int nch = (p.Cin >> 6) / SPLITK, chunk0 = zk * nch;      // this slice's channel chunks
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- halo pieces of this wave: piece wid + 8 q covers halo slots 8 piece .. + 7 (lane >> 3 = slot, lane & 7 = 16-byte chunk);
  // slot hp = 100 img + 10 hy + hx holds input pixel (hy - 1, hx - 1) of image (m0 / 64 + img); slots >= 400: nothing ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  const unsigned ldcb = (unsigned)p.ldc * 2u;
  unsigned h_off[NHQ];
#pragma unroll
  for (int q = 0; q < NHQ; ++q) {
    const int hp = (wid + 8 * q) * 8 + srow;
    const int img = hp / 100, rem = hp - img * 100, hy = rem / 10, hx = rem - hy * 10;
    const int iy = hy - 1, ix = hx - 1;
    const bool ok = hp < 400 && (unsigned)iy < 8u && (unsigned)ix < 8u;
    h_off[q] = ok ? (unsigned)(((m0 >> 6) + img) * 64 + iy * 8 + ix) * ldcb + dchunk * 16u : 0xFFFFFFFFu;
  }
  // weight pieces: piece wid (+ 8 for waves 0, 1) = rows 8 piece .. + 7 of the 80-row tile
  const int nwq = wid < WPIECES - 8 ? 2 : 1;
  unsigned w_off[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) w_off[q] = (unsigned)((n0 + (wid + 8 * q) * 8 + srow) * p.ldw * 2) + dchunk * 16u;

  // step s (global over the slice) = (chunk s / 9, tap s % 9); weights of a step: K offset (tap * Cin + 64 chunk) elements
  const int nsteps = nch * 9;
  auto stage_w = [&](int s, int slot) {
    const int c = s / 9, tap = s - 9 * c;
    const bool live = s < nsteps;
    const unsigned k0b = (unsigned)(tap * p.Cin + (chunk0 + c) * 64) * 2u;
    lds_dma16(rs_w, smem + W_BASE + slot * WBYTES + wid * 1024, live ? w_off[0] : 0xFFFFFFFFu, k0b);
    if (wid < WPIECES - 8) lds_dma16(rs_w, smem + W_BASE + slot * WBYTES + (wid + 8) * 1024, live ? w_off[1] : 0xFFFFFFFFu, k0b);
  };
  auto stage_h = [&](auto qc, int hbuf, int c) {
    constexpr int q = decltype(qc)::value;
    const bool live = c < nch;
    lds_dma16(rs_x, smem + hbuf * HBUF + (wid + 8 * q) * 1024, live ? h_off[q] : 0xFFFFFFFFu, (unsigned)(chunk0 + c) * 128u);
  };

  // ---- fragment addressing ----
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)(g ^ (lane & 7)) * 16u, fch1 = (unsigned)((g + 4) ^ (lane & 7)) * 16u;
  const unsigned w_base = (unsigned)(l15 * 128);
  int xhp[MI];   // halo slot of tap (0, 0) for this lane's output pixel of row block j: tile row t = wid * 32 + j * 16 + l15
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int t = wid * 32 + j * 16 + l15;
    xhp[j] = (t >> 6) * 100 + ((t >> 3) & 7) * 10 + (t & 7);
  }
  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: halo of the first chunk, weights of steps 0 .. D - 1 ----
  static_for<0, NHQ>([&](auto qc) { stage_h(qc, 0, 0); });
#pragma unroll
  for (int s = 0; s < D; ++s) stage_w(s, s);

  int hb = 0, slot = 0;
  for (int c = 0; c < nch; ++c) {
    static_for<0, 9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      // vmcnt: the weights of this step and (tap 0) every halo piece of this chunk have landed.  Younger than this step's
      // weights: the weights of the next D - 1 steps and the halo pieces the last three steps issued behind their weights
      // (taps 0-6 issue one each; tap 0 itself needs the one tap 6 issued: counted out)
      constexpr int hyoung = tap == 0 ? 0 : (tap == 1 ? 1 : (tap == 2 ? 2 : (tap <= 7 ? 3 : 2)));
      if (nwq == 2) wait_vm<2 * (D - 1) + hyoung>(); else wait_vm<(D - 1) + hyoung>();
      __builtin_amdgcn_s_barrier();
      // ---- operand addresses of this tap ----
      constexpr int ky = tap / 3, kx = tap - 3 * ky, tapoff = ky * 10 + kx;
      unsigned xa[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const unsigned hp = (unsigned)(xhp[j] + tapoff);
        xa[j] = lds0 + (unsigned)(hb * HBUF) + (hp << 7) + ((((unsigned)g ^ hp) & 7u) << 4);
      }
      const unsigned wa0 = lds0 + (unsigned)(W_BASE + slot * WBYTES) + w_base + fch0;
      const unsigned wa1 = lds0 + (unsigned)(W_BASE + slot * WBYTES) + w_base + fch1;
      u32x4 xf[MI][2], wf[NI][2];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        xf[j][0] = lds_read128(xa[j]);
        xf[j][1] = lds_read128(xa[j] ^ 64u);
      }
      auto rd = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        wf[i][0] = lds_read128o<i * 2048>(wa0);
        wf[i][1] = lds_read128o<i * 2048>(wa1);
      };
      rd(std::integral_constant<int, 0>{});
      rd(std::integral_constant<int, 1>{});
      rd(std::integral_constant<int, 2>{});
      wait_lgkm4<6>(xf[0][0], xf[0][1], xf[1][0], xf[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NI>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int issued = (i + 3) < NI ? (i + 3) : NI;
        wait_lgkm2<2 * (issued - i - 1)>(wf[i][0], wf[i][1]);
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          acc[i][j] = mma(wf[i][0], xf[j][0], acc[i][j]);
          acc[i][j] = mma(wf[i][1], xf[j][1], acc[i][j]);
          asm volatile("" : "+v"(acc[i][j]));
        }
        if constexpr (i + 3 < NI) rd(std::integral_constant<int, i + 3>{});
        if constexpr (i == 1) {
          // the slot the previous step left: weights of step + D; then (taps 0-6) one halo piece of the next chunk
          stage_w(c * 9 + tap + D, (slot + D) & (NS - 1));
          if constexpr (tap < NHQ) stage_h(std::integral_constant<int, tap>{}, hb ^ 1, c + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      slot = (slot + 1) & (NS - 1);
    });
    hb ^= 1;
  }
  wait_vm<0>();   // (dead pieces of the steps past the end: none may land after the workgroup has gone)

  // ---- fp32 slab of this K slice: a lane holds 4 consecutive columns of one row ----
  float* slab = reinterpret_cast<float*>(p.ws) + (long)zk * p.M * p.N;
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int m = m0 + wid * 32 + j * 16 + l15;
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n0 + i * 16 + 4 * g) = acc[i][j];
  }
}
}  // namespace cs8

// does the eight-wave 8 x 8-map kernel take this (bf16) convolution?  It always slices K four ways.
bool af_conv_s8_ok(const ConvGemmParams& p, int batch) {
  return batch == 1 && !p.fp8 && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.up == 0 && p.Ho == 8 && p.Wo == 8 && p.Hi == 8 && p.Wi == 8 &&
         p.Hs == 8 && p.Ws == 8 && p.M > 0 && p.M % cs8::BM == 0 && p.N % cs8::BN == 0 && p.Cin % (64 * cs8::SPLITK) == 0 && p.K == 9 * p.Cin &&
         p.ldc >= p.Cin && p.ldc % 8 == 0 && p.epilogue == AF_EPI_NONE && !p.ln_stats && !p.ln_stats_out && !p.gn_ab && !p.phase4 &&
         p.src_batch_stride == (long)64 * p.ldc;
}
int af_launch_conv_s8(const ConvGemmParams& p, hipStream_t stream) {
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&cs8::conv3x3_s8_kernel), cs8::LDS_BYTES)) return rc;
  if (p.splitk != cs8::SPLITK || !p.ws) { af_set_error_msg("conv_s8: needs the four-slice slab workspace"); return -1; }
  hipLaunchKernelGGL(cs8::conv3x3_s8_kernel, dim3((p.M / cs8::BM) * (p.N / cs8::BN), 1, cs8::SPLITK), dim3(512), cs8::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
