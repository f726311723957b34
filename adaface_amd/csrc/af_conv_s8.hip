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
//
// The same kernel with ONE K slice runs the 16 x 16 maps (1280 -> 1280 x11, 2560 -> 1280 x2, 640 -> 1280 x1 per forward; M = 4096):
// a tile is one whole image x 80 columns -- 16 x 16 = 256 tiles, so no K slices, no slabs, no reduce launch (round 3: 256 x 160
// tiles over two slices + reduce) -- its 18 x 18 halo (324 pixels) resident per chunk; bias, time-embedding row and residual are
// applied on the accumulators and the bf16 rows leave through a wave-private transposition tile as 16-byte stores.
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

template <int SPLITK_T>
__global__ __launch_bounds__(512) void conv3x3_s8_kernel(const ConvGemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int ntn = p.N / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;   // (consecutive ids share the row tile = its halo: same XCD or not, L2 or MALL)
  const int m0 = tm * BM, n0 = tn * BN, zk = blockIdx.z;
  const int nch = (p.Cin >> 6) / SPLITK_T, chunk0 = zk * nch;    // this slice's channel chunks
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- halo pieces of this wave: piece wid + 8 q covers halo slots 8 piece .. + 7 (lane >> 3 = slot, lane & 7 = 16-byte chunk);
  // slot hp = HP1 img + HW hy + hx holds input pixel (hy - 1, hx - 1) of image (img0 + img); slots past the last image: nothing ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  const unsigned ldcb = (unsigned)p.ldc * 2u;
  // geometry: a tile = 256 consecutive pixels in (image, y, x) order = `ipt` whole images of Wo x Wo (Wo = 8: four, 16: one) or R = 256 / Wo
  // whole rows of one image (Wo = 32, 64).  Halo of one image part: (R + 2) rows x HW = Wo + 2 columns = HP1 slots
  const int wsh = p.wo_shift, Wo = 1 << wsh, HW = Wo + 2, hwsh = p.howo_shift;
  const int R = min(p.Ho, 256 >> wsh), HP1 = (R + 2) * HW, ipt = 256 / (R << wsh);
  const int img0 = m0 >> hwsh, y0 = (m0 & ((1 << hwsh) - 1)) >> wsh;
  unsigned h_off[NHQ];
#pragma unroll
  for (int q = 0; q < NHQ; ++q) {
    const int hp = (wid + 8 * q) * 8 + srow;
    const int img = hp / HP1, rem = hp - img * HP1, hy = rem / HW, hx = rem - hy * HW;
    const int iy = y0 + hy - 1, ix = hx - 1;
    const bool ok = img < ipt && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)Wo;
    h_off[q] = ok ? (unsigned)(((img0 + img) << hwsh) + (iy << wsh) + ix) * ldcb + dchunk * 16u : 0xFFFFFFFFu;
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
    const int part = t / (R << wsh), yl = (t >> wsh) - part * R;
    xhp[j] = part * HP1 + yl * HW + (t & (Wo - 1));
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

  // Software pipeline over the tap steps: the MFMAs of the last two weight blocks of a step (8 of its 20) are HELD BACK and
  // issued behind the next step's barrier, while that step's first fragment reads are in flight -- a wave otherwise sits out the
  // LDS latency after every barrier with nothing to give the matrix pipe.  Two register sets (activation fragments + the
  // two held weight blocks) alternate with the tap's parity; nine taps per chunk: one copy of the set per chunk.
  u32x4 xs[2][MI][2], wh[2][2][2];
#pragma unroll
  for (int j = 0; j < MI; ++j) xs[1][j][0] = xs[1][j][1] = u32x4{0u, 0u, 0u, 0u};     // (the first step multiplies zeros)
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2) wh[1][b2][0] = wh[1][b2][1] = u32x4{0u, 0u, 0u, 0u};
  int hb = 0, slot = 0;
  for (int c = 0; c < nch; ++c) {
    static_for<0, 9>([&](auto tc) {
      constexpr int tap = decltype(tc)::value;
      constexpr int cs = tap & 1, ps = cs ^ 1;      // this step's register set / the previous step's
      // vmcnt: the weights of this step and (tap 0) every halo piece of this chunk have landed.  Younger than this step's
      // weights: the weights of the next D - 1 steps and the halo pieces the last three steps issued behind their weights
      // (taps 0-6 issue one each; tap 0 itself needs the one tap 6 issued: counted out)
      constexpr int hyoung = tap == 0 ? 0 : (tap == 1 ? 1 : (tap == 2 ? 2 : (tap <= 7 ? 3 : 2)));
      if (nwq == 2) wait_vm<2 * (D - 1) + hyoung>(); else wait_vm<(D - 1) + hyoung>();
      __builtin_amdgcn_s_barrier();
      // ---- operand addresses of this tap ----
      constexpr int ky = tap / 3, kx = tap - 3 * ky;
      const int tapoff = ky * HW + kx;
      unsigned xa[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const unsigned hp = (unsigned)(xhp[j] + tapoff);
        xa[j] = lds0 + (unsigned)(hb * HBUF) + (hp << 7) + ((((unsigned)g ^ hp) & 7u) << 4);
      }
      const unsigned wa0 = lds0 + (unsigned)(W_BASE + slot * WBYTES) + w_base + fch0;
      const unsigned wa1 = lds0 + (unsigned)(W_BASE + slot * WBYTES) + w_base + fch1;
      u32x4 wf[3][2];
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        xs[cs][j][0] = lds_read128(xa[j]);
        xs[cs][j][1] = lds_read128(xa[j] ^ 64u);
      }
      wf[0][0] = lds_read128o<0 * 2048>(wa0); wf[0][1] = lds_read128o<0 * 2048>(wa1);
      wf[1][0] = lds_read128o<1 * 2048>(wa0); wf[1][1] = lds_read128o<1 * 2048>(wa1);
      wf[2][0] = lds_read128o<2 * 2048>(wa0); wf[2][1] = lds_read128o<2 * 2048>(wa1);
      __builtin_amdgcn_sched_barrier(0);
      // ---- the previous step's blocks 3, 4 (registers only) ----
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
          acc[3 + b2][j] = mma(wh[ps][b2][0], xs[ps][j][0], acc[3 + b2][j]);
          acc[3 + b2][j] = mma(wh[ps][b2][1], xs[ps][j][1], acc[3 + b2][j]);
          asm volatile("" : "+v"(acc[3 + b2][j]));
        }
      __builtin_amdgcn_sched_barrier(0);
      // ---- blocks 0, 1, 2 of this step; blocks 3, 4 are read for the next one ----
      wait_lgkm4<4>(xs[cs][0][0], xs[cs][0][1], xs[cs][1][0], xs[cs][1][1]);
      wait_lgkm2<4>(wf[0][0], wf[0][1]);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        acc[0][j] = mma(wf[0][0], xs[cs][j][0], acc[0][j]);
        acc[0][j] = mma(wf[0][1], xs[cs][j][1], acc[0][j]);
        asm volatile("" : "+v"(acc[0][j]));
      }
      wh[cs][0][0] = lds_read128o<3 * 2048>(wa0); wh[cs][0][1] = lds_read128o<3 * 2048>(wa1);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm2<4>(wf[1][0], wf[1][1]);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        acc[1][j] = mma(wf[1][0], xs[cs][j][0], acc[1][j]);
        acc[1][j] = mma(wf[1][1], xs[cs][j][1], acc[1][j]);
        asm volatile("" : "+v"(acc[1][j]));
      }
      // the slot the previous step left: weights of step + D; then (taps 0-6) one halo piece of the next chunk
      stage_w(c * 9 + tap + D, (slot + D) & (NS - 1));
      if constexpr (tap < NHQ) stage_h(std::integral_constant<int, tap>{}, hb ^ 1, c + 1);
      wh[cs][1][0] = lds_read128o<4 * 2048>(wa0); wh[cs][1][1] = lds_read128o<4 * 2048>(wa1);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm2<4>(wf[2][0], wf[2][1]);
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        acc[2][j] = mma(wf[2][0], xs[cs][j][0], acc[2][j]);
        acc[2][j] = mma(wf[2][1], xs[cs][j][1], acc[2][j]);
        asm volatile("" : "+v"(acc[2][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
      // every read of this step's slot is back before the wave reaches the next barrier (the slot is restaged behind it)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(wh[cs][0][0]), "+v"(wh[cs][0][1]), "+v"(wh[cs][1][0]), "+v"(wh[cs][1][1])
                   :
                   : "memory");
      slot = (slot + 1) & (NS - 1);
    });
    // tap 8 left its fragments in set 0; tap 0 of the next chunk looks for them in set 1
#pragma unroll
    for (int j = 0; j < MI; ++j) { xs[1][j][0] = xs[0][j][0]; xs[1][j][1] = xs[0][j][1]; }
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) { wh[1][b2][0] = wh[0][b2][0]; wh[1][b2][1] = wh[0][b2][1]; }
    hb ^= 1;
  }
  // the last step's held-back blocks
#pragma unroll
  for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      acc[3 + b2][j] = mma(wh[1][b2][0], xs[1][j][0], acc[3 + b2][j]);
      acc[3 + b2][j] = mma(wh[1][b2][1], xs[1][j][1], acc[3 + b2][j]);
    }
  wait_vm<0>();   // (dead pieces of the steps past the end: none may land after the workgroup has gone)

  if constexpr (SPLITK_T > 1) {
    // ---- fp32 slab of this K slice: a lane holds 4 consecutive columns of one row ----
    float* slab = reinterpret_cast<float*>(p.ws) + (long)zk * p.M * p.N;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int m = m0 + wid * 32 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n0 + i * 16 + 4 * g) = acc[i][j];
    }
  } else {
    // ---- one K slice: bias + time-embedding row + residual on the accumulators, bf16 through a wave-private transposition tile
    // (where the halo buffers were: every wave is past its last fragment read) into 16-byte row stores ----
    __builtin_amdgcn_s_barrier();
    constexpr int OPITCH = BN * 2 + 16, OCH = BN / 8, NST = 32 * OCH / 64;     // 176-byte rows, 10 chunks per row, 5 stores per lane
    char* otile = smem + wid * (32 * OPITCH);
    const T* res = reinterpret_cast<const T*>(p.residual);
    const T* rowb = reinterpret_cast<const T*>(p.rowbias);
    const int cl = 4 * g;
    float4 bvec[NI];
    Quad<T> bq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bvec[i] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + i * 16 + cl) : float4{0.f, 0.f, 0.f, 0.f};
      if (rowb) bq[i].load(rowb + (long)(m0 >> p.howo_shift) * p.ldrb + n0 + i * 16 + cl);   // (a tile lies inside one sample)
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int m = m0 + wid * 32 + j * 16 + l15;
      Quad<T> rq[NI];
      if (res) {
#pragma unroll
        for (int i = 0; i < NI; ++i) rq[i].load(res + (long)m * p.ldr + n0 + i * 16 + cl);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float* bp = reinterpret_cast<const float*>(&bvec[i]);
        Quad<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e] + bp[e];
          if (rowb) v += to_f32<T>(bq[i].e[e]);
          if (res) v += to_f32<T>(rq[i].e[e]);
          o.e[e] = from_f32<T>(v);
        }
        o.store(reinterpret_cast<T*>(otile + (j * 16 + l15) * OPITCH) + i * 16 + cl);
      }
    }
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(p.out), 0, (int)0xFFFFFFF0u, 0x00020000);
    u32x4 chunk[NST];
    unsigned orow[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const unsigned c = (unsigned)(lane + 64 * i), row = c / OCH, ch = c - row * OCH;
      orow[i] = (unsigned)(m0 + wid * 32 + (int)row) * (unsigned)(p.ldo * 2) + ch * 16u;
      chunk[i] = *reinterpret_cast<const u32x4*>(otile + row * OPITCH + ch * 16);
    }
    __builtin_amdgcn_sched_barrier(0);   // (all chunks and offsets first, then the stores back to back: scripts/check_isa_hazards.py)
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(chunk[i], rs_o, orow[i], n0 * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
}  // namespace cs8

// does the small-map kernel take this (bf16) convolution, and in how many K slices?  8 x 8 maps: 4; 16 x 16 maps: 1; 0 = no
int af_conv_s8_slices(const ConvGemmParams& p, int batch) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  if (batch != 1 || p.fp8 || p.ks != 3 || p.stride != 1 || p.pad != 1 || p.up != 0 || p.Hi != p.Ho || p.Wi != p.Wo ||
      p.Hs != p.Ho || p.Ws != p.Wo || !pow2(p.Wo) || !pow2(p.Ho) || p.Wo < 8 || p.Wo > 64 || p.M <= 0 || p.M % cs8::BM != 0 || p.N % cs8::BN != 0 ||
      p.K != 9 * p.Cin || p.ldc < p.Cin || p.ldc % 8 != 0 || p.epilogue != AF_EPI_NONE || p.ln_stats || p.ln_stats_out || p.gn_ab || p.phase4 ||
      p.src_batch_stride != (long)p.Ho * p.Wo * p.ldc)
    return 0;
  // the tile's halo must fit the buffer: whole images (Ho * Wo <= 256) or 256 / Wo whole rows of one image
  const int R = p.Ho < 256 / p.Wo ? p.Ho : 256 / p.Wo;
  if (256 % (R * p.Wo) != 0 || (256 / (R * p.Wo)) * (R + 2) * (p.Wo + 2) > cs8::HSLOTS || (p.Ho * p.Wo > 256 && p.Ho % R != 0)) return 0;
  if (p.Wo == 8 && p.Ho == 8) return p.Cin % (64 * cs8::SPLITK) == 0 ? cs8::SPLITK : 0;
  return (p.Cin % 64 == 0 && p.ldo % 8 == 0 && (!p.residual || p.ldr % 4 == 0) && (!p.rowbias || p.ldrb % 4 == 0) && ((__UINTPTR_TYPE__)p.out & 15) == 0 &&
          (long)(p.M / cs8::BM) * (p.N / cs8::BN) >= 128) ? 1 : 0;
}
bool af_conv_s8_ok(const ConvGemmParams& p, int batch) { return af_conv_s8_slices(p, batch) != 0; }
int af_launch_conv_s8(const ConvGemmParams& p, hipStream_t stream) {
  static unsigned long long attr_done4 = 0, attr_done1 = 0;
  const int sl = af_conv_s8_slices(p, 1);
  if (sl == 0 || p.splitk != sl || (sl > 1 && !p.ws) || p.wo_shift < 3 || p.howo_shift < p.wo_shift) {
    af_set_error_msg("conv_s8: shape / slice count %d not taken by the small-map kernel", p.splitk);
    return -1;
  }
  const dim3 grid((p.M / cs8::BM) * (p.N / cs8::BN), 1, sl);
  if (sl > 1) {
    if (int rc = af_ensure_dynamic_lds(attr_done4, reinterpret_cast<const void*>(&cs8::conv3x3_s8_kernel<cs8::SPLITK>), cs8::LDS_BYTES)) return rc;
    hipLaunchKernelGGL(cs8::conv3x3_s8_kernel<cs8::SPLITK>, grid, dim3(512), cs8::LDS_BYTES, stream, p);
  } else {
    if (int rc = af_ensure_dynamic_lds(attr_done1, reinterpret_cast<const void*>(&cs8::conv3x3_s8_kernel<1>), cs8::LDS_BYTES)) return rc;
    hipLaunchKernelGGL(cs8::conv3x3_s8_kernel<1>, grid, dim3(512), cs8::LDS_BYTES, stream, p);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
