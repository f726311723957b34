// Standalone lab for the 8-wave "ping-pong" GEMM main loop (not part of the shipped library).
//   C[M,N] (bf16) = X[M,K] (bf16) * W[N,K]^T (bf16), fp32 accumulate.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_pp_lab gemm_pp_lab.hip ; run on the GPU box.
//
// Structure under test: 256 x BN x 64 tile, 512 threads.  Waves 0-3 (group 0, one per SIMD) and waves 4-7 (group 1,
// their SIMD partners) run the same program one barrier apart: while one group issues its MFMAs for K-tile t the
// other reads its fragments of the next tile from LDS and issues LDS-DMA for the tile two ahead.  LDS holds a ring
// of three K-tiles filled by buffer_load...lds with counted vmcnt, raw s_barrier (no __syncthreads, no vmcnt(0)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <cstdint>
#include <type_traits>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}

__device__ __forceinline__ void tile_coords(int wg, int nwg, int ntm, int ntn, int gm, int& tm, int& tn) {
  const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  const int per_group = gm * ntn;
  const int gid = wg / per_group;
  const int first_m = gid * gm;
  const int gsz = min(ntm - first_m, gm);
  const int in_g = wg - gid * per_group;
  tn = in_g / gsz;
  tm = first_m + (in_g - tn * gsz);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// LDS fragment read the compiler cannot see as an LDS access: hipcc otherwise puts s_waitcnt vmcnt(0) in front of the
// first ds_read after any LDS-DMA, which drains the ring every K-tile.  Ordering is by hand (wait_lgkm0 + sched_barrier).
template <int OFF> __device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// MODE bits: 1 = group 1 issues its DMA right after the barrier that opens its compute segment (one tile further ahead)
//            2 = balanced DMA split (7 / 6 pieces per wave instead of 8 / 5)
//            4 = ABLATION: no DMA inside the loop (wrong results)      8 = ABLATION: no ds_reads inside the loop (wrong results)
//           16 = DMA pieces interleaved between the MFMAs of the compute segment (both groups)
template <int BN, int MODE>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W,
                                                      bf16* __restrict__ C, int M, int N, int K, int gm,
                                                      unsigned long long* __restrict__ dbg, int stagger_units = 0) {
  constexpr int BM = 256, BK = 64;
  constexpr bool STAMP = (MODE & 32) != 0;
  constexpr bool PERSIST = (MODE & 128) != 0;
  constexpr bool STAGGER = (MODE & 256) != 0;   // persistent workgroups start a quarter tile apart (de-synchronises the output bursts)   // workgroups loop over output tiles; the DMA ring runs across tile boundaries
  constexpr int HN = BN / 2;          // columns per wave group
  constexpr int NI = HN / 16;         // 16-wide n tiles per wave
  constexpr int MI = 4;               // 16-high m tiles per wave (64 rows)
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, SLOT = XBYTES + WBYTES;
  constexpr int XP = BM / 8, WP = BN / 8;                 // 1-KiB pieces (8 rows x 128 B) per K-tile: 32 + 20
  constexpr bool BAL = (MODE & 2) != 0;
  constexpr int XP0 = BAL ? ((XP + WP + 7) / 8) * 4 : XP; // X pieces staged by group 0 (4 waves)
  constexpr int NP0 = XP0 / 4;                            // pieces per wave, group 0
  constexpr int XP1 = XP - XP0;                           // X pieces left to group 1
  constexpr int NP1 = XP1 / 4 + WP / 4;                   // pieces per wave, group 1
  constexpr int NPMAX = NP0 > NP1 ? NP0 : NP1;
  constexpr bool EARLY = (MODE & 1) != 0 || (MODE & 16) != 0;
  constexpr bool INTER = (MODE & 16) != 0;
  constexpr bool NO_DMA = (MODE & 4) != 0, NO_DS = (MODE & 8) != 0;
  static_assert(XP0 % 4 == 0 && XP1 % 4 == 0 && WP % 4 == 0, "piece split");
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  unsigned long long t_entry = 0;
  if constexpr ((MODE & 32) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry)::"memory");
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wid >> 2, wq = wid & 3;
  const int ntm = (M + BM - 1) / BM, ntn = N / BN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, ntm, ntn, gm, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = K / BK;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- staging: piece q of this wave = 8 rows x 128 B = one LDS-DMA instruction ----
  //   group 0 wave wq: X pieces wq + 4q, q < NP0
  //   group 1 wave wq: X pieces XP0 + wq + 4q, q < XP1/4, then W pieces wq + 4(q - XP1/4)
  const int srow = lane >> 3;
  const unsigned lchunk = (unsigned)((lane & 7) ^ srow);  // XOR swizzle applied on the SOURCE address
  unsigned goff[NPMAX];
  auto compute_goff = [&](int m0_, int n0_) {
    static_for<0, NPMAX>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      unsigned v = 0xFFFFFFFFu;
      if (g == 0) {
        if (q < NP0) { const int m = m0_ + (wq + 4 * q) * 8 + srow; if (m < M) v = (unsigned)m * (unsigned)K * 2u + lchunk * 16u; }
      } else if (q < XP1 / 4) {
        const int m = m0_ + (XP0 + wq + 4 * q) * 8 + srow; if (m < M) v = (unsigned)m * (unsigned)K * 2u + lchunk * 16u;
      } else if (q < NP1) {
        v = (unsigned)(n0_ + (wq + 4 * (q - XP1 / 4)) * 8 + srow) * (unsigned)K * 2u + lchunk * 16u;
      }
      goff[q] = v;
    });
  };
  compute_goff(m0, n0);
  auto piece = [&](auto qc, int slot_off, unsigned kb) {
    constexpr int q = decltype(qc)::value;
    if (g == 0) {
      if constexpr (q < NP0) lds_dma16(rs_x, smem + slot_off + (wq + 4 * q) * 1024, goff[q], kb);
    } else {
      if constexpr (q < XP1 / 4) lds_dma16(rs_x, smem + slot_off + (XP0 + wq + 4 * q) * 1024, goff[q], kb);
      else if constexpr (q < NP1) lds_dma16(rs_w, smem + slot_off + XBYTES + (wq + 4 * (q - XP1 / 4)) * 1024, goff[q], kb);
    }
  };
  auto stage = [&](int slot_off, int kt) {
    const unsigned kb = (unsigned)kt * 128u;
    static_for<0, NPMAX>([&](auto qc) { piece(qc, slot_off, kb); });
  };
  // counted waits: "all but the newest tile's pieces of this wave have landed"
  auto wait_keep1 = [&]() { if (g == 0) wait_vm<NP0>(); else wait_vm<NP1>(); };
  auto wait_keep2 = [&]() { if (g == 0) wait_vm<2 * NP0>(); else wait_vm<2 * NP1>(); };

  // ---- fragment addresses: row (lane & 15) of a 16-row tile, chunk ((lane >> 4) + 4 s) ^ (row & 7) ----
  const int frow = lane & 15;
  const unsigned fch0 = (unsigned)(((lane >> 4)) ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)(((lane >> 4) + 4) ^ (lane & 7)) * 16u;
  const unsigned x_base = (unsigned)((wq * 64 + frow) * 128);
  const unsigned w_base = (unsigned)(XBYTES + (g * HN + frow) * 128);

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 xf[MI][2], wf[NI][2];
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  auto load_frags = [&](int slot_off) {
    const unsigned s = lds0 + (unsigned)slot_off;
    const unsigned w0 = s + w_base + fch0, w1 = s + w_base + fch1, x0 = s + x_base + fch0, x1 = s + x_base + fch1;
    static_for<0, NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wf[i][0] = lds_read128<i * 2048>(w0);
      wf[i][1] = lds_read128<i * 2048>(w1);
    });
    static_for<0, MI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      xf[j][0] = lds_read128<j * 2048>(x0);
      xf[j][1] = lds_read128<j * 2048>(x1);
    });
  };
  auto read_unit = [&](int slot_off, auto sc) {
    constexpr int u = decltype(sc)::value;
    const unsigned b = lds0 + (unsigned)slot_off + (u ? fch1 : fch0);
    static_for<0, NI>([&](auto ic) { constexpr int i = decltype(ic)::value; wf[i][u] = lds_read128<i * 2048>(b + w_base); });
    static_for<0, MI>([&](auto jc) { constexpr int j = decltype(jc)::value; xf[j][u] = lds_read128<j * 2048>(b + x_base); });
  };
  auto mfma_unit = [&](auto sc) {
    constexpr int u = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][u]),
                                                            __builtin_bit_cast(bf16x8, xf[j][u]), acc[i][j], 0, 0, 0);
  };
  // one half of a lag-pipeline compute segment: the nine fragment reads of unit UR go out one per MFMA behind the
  // first nine MFMAs of unit UM (in-order issue: a read placed after the MFMAs would not overlap them)
  auto half = [&](int slot_off, auto rc, auto mc) {
    constexpr int ur = decltype(rc)::value, um = decltype(mc)::value;
    const unsigned b = lds0 + (unsigned)slot_off + (ur ? fch1 : fch0);
    static_for<0, NI * MI>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int i = n / MI, j = n % MI;
      if constexpr (n < NI) wf[n][ur] = lds_read128<n * 2048>(b + w_base);
      else if constexpr (n < NI + MI) xf[n - NI][ur] = lds_read128<(n - NI) * 2048>(b + x_base);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][um]),
                                                          __builtin_bit_cast(bf16x8, xf[j][um]), acc[i][j], 0, 0, 0);
      if constexpr (n < NI + MI) __builtin_amdgcn_sched_barrier(0);
    });
  };
  // MFMAs of one K-tile; with INTER, one DMA piece goes out after every (NMF / NPMAX) MFMAs
  auto compute = [&](bool do_stage, int slot_off, unsigned kb) {
    constexpr int NMF = 2 * NI * MI;
    constexpr int EVERY = NMF / (NPMAX + 1);
    static_for<0, NMF>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int s = n / (NI * MI), i = (n / MI) % NI, j = n % MI;
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][s]),
                                                          __builtin_bit_cast(bf16x8, xf[j][s]), acc[i][j], 0, 0, 0);
      if constexpr (INTER && (n % EVERY) == EVERY - 1 && (n / EVERY) < NPMAX) {
        if (do_stage) piece(std::integral_constant<int, n / EVERY>{}, slot_off, kb);
      }
    });
  };

  // ---- epilogue: acc[i][j][r] -> C[m0 + wq*64 + j*16 + (lane&15)][n0 + g*HN + i*16 + 4*(lane>>4) + r] ----
  auto store_tile = [&](int m0_, int n0_) {
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int m = m0_ + wq * 64 + j * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int n = n0_ + g * HN + i * 16 + 4 * (lane >> 4);
        union { bf16 h[4]; uint2 u; } pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) pk.h[r] = (bf16)acc[i][j][r];
        *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = pk.u;
      }
    }
  };
  constexpr bool LAG = (MODE & 64) != 0;
  std::integral_constant<int, 0> U0; std::integral_constant<int, 1> U1;
  auto stamp = [&]() -> unsigned long long {
    unsigned long long v = 0;
    if constexpr (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    return v;
  };
  unsigned long long s_load = 0, s_b1 = 0, s_comp = 0, s_vm = 0, s_b2 = 0, rt0 = 0, tbeg = 0, ta = 0;
  if constexpr (PERSIST) {
    const int ntiles = ntm * ntn;
    const int nloc = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup
    const int G = nloc * KT;                                                             // its K-tile stream
    auto tile_of = [&](int i, int& m0_, int& n0_) {
      int tm_, tn_;
      tile_coords((int)blockIdx.x + i * (int)gridDim.x, ntiles, ntm, ntn, gm, tm_, tn_);
      m0_ = tm_ * BM; n0_ = tn_ * BN;
    };
    if constexpr (STAGGER) {
      const int phase = ((int)blockIdx.x >> 3) & 3;                 // blocks b, b+8, ... share an XCD
      for (int i = 0; i < phase * stagger_units; ++i) __builtin_amdgcn_s_sleep(32);   // 32 x 64 cycles each
    }
    int st_i = 0, st_kt = 0;          // staging cursor (runs 1-2 K tiles ahead of the compute cursor)
    { int a_, b_; tile_of(0, a_, b_); compute_goff(a_, b_); }
    auto stage_next = [&](int slot_off) {
      stage(slot_off, st_kt);
      if (++st_kt == KT) {
        st_kt = 0;
        if (++st_i < nloc) { int a_, b_; tile_of(st_i, a_, b_); compute_goff(a_, b_); }
      }
    };
    int cm0, cn0, c_i = 0, c_kt = 0;  // compute cursor
    tile_of(0, cm0, cn0);
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < MI; ++j) xf[j][1] = u32x4{0u, 0u, 0u, 0u};
    stage_next(0);
    if (g == 1) { if (G > 1) { stage_next(SLOT); wait_keep1(); } else wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();
    int rd = 0, w0 = SLOT, w1 = 2 * SLOT;
    unsigned long long s_epi = 0;
    if constexpr (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
    tbeg = stamp(); ta = tbeg;
    for (int t = 0; t < G; ++t) {
      if (g == 0) { if (t + 1 < G) { stage_next(w0); wait_keep1(); } else wait_vm<0>(); }
      else { if (t + 2 < G) stage_next(w1); }
      const unsigned long long tb = stamp();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tc = stamp();
      __builtin_amdgcn_s_setprio(1);
      half(rd, U0, U1);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      half(rd, U1, U0);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      const unsigned long long td = stamp();
      if (++c_kt == KT) {
        // tile finished: the trailing half tile, the output, then a clean accumulator / "unit -1" for the next tile
        mfma_unit(U1);
        store_tile(cm0, cn0);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < MI; ++j) xf[j][1] = u32x4{0u, 0u, 0u, 0u};
        c_kt = 0;
        if (++c_i < nloc) tile_of(c_i, cm0, cn0);
      }
      const unsigned long long tx = stamp();
      if (g == 1) { if (t + 2 < G) wait_keep1(); else wait_vm<0>(); }
      const unsigned long long te = stamp();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tf = stamp();
      if constexpr (STAMP) { s_load += tb - ta; s_b1 += tc - tb; s_comp += td - tc; s_epi += tx - td; s_vm += te - tx; s_b2 += tf - te; ta = tf; }
      const int tmp = rd; rd = w0; w0 = w1; w1 = tmp;
    }
    if constexpr (STAMP) {
      const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
      if (blockIdx.x == 0 && lane == 0) {
        unsigned long long* d = dbg + wid * 8;
        d[0] = s_load; d[1] = s_b1; d[2] = s_comp; d[3] = s_vm; d[4] = s_b2; d[5] = ta - tbeg; d[6] = rt1 - rt0; d[7] = G;
        dbg[64 + wid] = s_epi;
      }
    }
    if (g == 0) __builtin_amdgcn_s_barrier();
  } else
  if constexpr (LAG) {
    // ---- "lag" pipeline: the DMA segment only issues DMA; the compute segment reads the K-halves (units) of tile t
    // while the MFMAs run half a tile behind: [read unit 2t | MFMA unit 2t-1] [read unit 2t+1 | MFMA unit 2t].
    // group 0 issues tile t+1 in D(t), group 1 issues tile t+2 in D(t).
#pragma unroll
    for (int i = 0; i < NI; ++i) wf[i][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < MI; ++j) xf[j][1] = u32x4{0u, 0u, 0u, 0u};
    stage(0, 0);
    if (g == 1) { if (KT > 1) { stage(SLOT, 1); wait_keep1(); } else wait_vm<0>(); }
    if constexpr (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
    tbeg = stamp(); ta = tbeg;
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();
    int rd = 0, w0 = SLOT, w1 = 2 * SLOT;   // slot of tile t, t+1, t+2
    for (int t = 0; t < KT; ++t) {
      // ---------------- DMA segment ----------------
      if (!NO_DMA) {
        if (g == 0) { if (t + 1 < KT) stage(w0, t + 1); } else { if (t + 2 < KT) stage(w1, t + 2); }
      }
      if (g == 0) { if (t + 1 < KT) wait_keep1(); else wait_vm<0>(); }   // own part of tile t landed
      const unsigned long long tb = stamp();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tc = stamp();
      // ---------------- compute segment ----------------
      __builtin_amdgcn_s_setprio(1);
      half(rd, U0, U1);                    // (t == 0: MFMAs on the zero-initialised fragments of unit -1)
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      half(rd, U1, U0);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
      const unsigned long long td = stamp();
      if (g == 1) { if (t + 2 < KT) wait_keep1(); else wait_vm<0>(); }   // own part of tile t+1 landed
      const unsigned long long te = stamp();
      __builtin_amdgcn_s_barrier();
      const unsigned long long tf = stamp();
      if constexpr (STAMP) { s_load += tb - ta; s_b1 += tc - tb; s_comp += td - tc; s_vm += te - td; s_b2 += tf - te; ta = tf; }
      const int tmp = rd; rd = w0; w0 = w1; w1 = tmp;
    }
    mfma_unit(U1);
    if (g == 0) __builtin_amdgcn_s_barrier();
  } else {
  // ---- prologue: tiles 0 and 1 in flight (EARLY: group 1 also tile 2), tile 0 landed ----
  stage(0, 0);
  if (KT > 1) stage(SLOT, 1);
  if (EARLY && g == 1 && KT > 2) stage(2 * SLOT, 2);
  if (EARLY && g == 1) { if (KT > 2) wait_keep2(); else if (KT > 1) wait_keep1(); else wait_vm<0>(); }
  else { if (KT > 1) wait_keep1(); else wait_vm<0>(); }
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();  // stagger group 1 by one interval

  if constexpr (STAMP) rt0 = __builtin_amdgcn_s_memrealtime();
  tbeg = stamp();
  ta = tbeg;
  int rd = 0, wr = 2 * SLOT;
  for (int t = 0; t < KT; ++t) {
    // ---------------- load segment: fragments of tile t (+ DMA for tile t+2) ----------------
    if (!NO_DS || t == 0) load_frags(rd);
    const bool more = t + 2 < KT;
    if (!INTER && !NO_DMA && (!EARLY || g == 0)) {
      if (more) stage(wr, t + 2);
    }
    // group 1: its part of tile t+1 must have landed before group 0 reads it in the next interval
    if (g == 1) { if (more) wait_keep1(); else wait_vm<0>(); }
    wait_lgkm0();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long tb = stamp();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tc = stamp();
    // ---------------- compute segment ----------------
    // group 1 stages tile t+3 into tile t's slot (every read of it completed before the barrier above);
    // with INTER group 0 stages tile t+2 here too (its slot was released one interval ago)
    const bool st1 = !NO_DMA && EARLY && g == 1 && t + 3 < KT;
    const bool st0 = !NO_DMA && INTER && g == 0 && more;
    if (!INTER && st1) stage(rd, t + 3);
    __builtin_amdgcn_s_setprio(1);
    compute(st0 || st1, g == 0 ? wr : rd, (unsigned)(g == 0 ? t + 2 : t + 3) * 128u);
    __builtin_amdgcn_s_setprio(0);
    const unsigned long long td = stamp();
    if (g == 0) { if (more) wait_keep1(); else wait_vm<0>(); }   // own part of tile t+1 landed before next interval's reads
    const unsigned long long te = stamp();
    __builtin_amdgcn_s_barrier();
    const unsigned long long tf = stamp();
    if constexpr (STAMP) { s_load += tb - ta; s_b1 += tc - tb; s_comp += td - tc; s_vm += te - td; s_b2 += tf - te; ta = tf; }
    rd = (rd == 2 * SLOT) ? 0 : rd + SLOT;
    wr = (wr == 2 * SLOT) ? 0 : wr + SLOT;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();
  }
  if constexpr (STAMP && !PERSIST) {
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && lane == 0) {
      unsigned long long* d = dbg + wid * 8;
      d[0] = s_load; d[1] = s_b1; d[2] = s_comp; d[3] = s_vm; d[4] = s_b2; d[5] = ta - tbeg; d[6] = rt1 - rt0; d[7] = KT;
    }
  }

  if constexpr (!PERSIST) store_tile(m0, n0);
  if constexpr (STAMP && !PERSIST) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores have left the wave
    unsigned long long t_exit;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_exit)::"memory");
    if (blockIdx.x == 0 && lane == 0) { dbg[80 + wid * 4 + 0] = tbeg - t_entry; dbg[80 + wid * 4 + 1] = ta - tbeg; dbg[80 + wid * 4 + 2] = t_exit - ta; }
  }
}

static uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static unsigned long long* g_dbg = nullptr;
static int g_stagger = 0;
template <int BN, int MODE>
static double run(const char* name, const bf16* dX, const bf16* dW, bf16* dC, int M, int N, int K, int gm, int iters) {
  if (!g_dbg) CK(hipMalloc(&g_dbg, 128 * 8));
  constexpr int SLOT = (256 + BN) * 128;
  const int lds = 3 * SLOT;
  CK(hipFuncSetAttribute((const void*)gemm_pp_kernel<BN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int ntm = (M + 255) / 256, ntn = N / BN;
  int nblk = ntm * ntn;
  if (MODE & 128) {   // persistent: equal share of tiles per workgroup, at most one workgroup per CU
    const int rounds = (nblk + 255) / 256;
    nblk = (nblk + rounds - 1) / rounds;
  }
  dim3 grid(nblk), block(512);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_pp_kernel<BN, MODE>), grid, block, lds, 0, dX, dW, dC, M, N, K, gm, g_dbg, g_stagger);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm_pp_kernel<BN, MODE>), grid, block, lds, 0, dX, dW, dC, M, N, K, gm, g_dbg, g_stagger);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / iters;
  const double tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
  printf("%-28s M=%6d N=%5d K=%5d grid=%4d  %8.1f us  %7.1f TF/s\n", name, M, N, K, ntm * ntn, us, tf);
  if (MODE & 32) {
    unsigned long long h[128];
    CK(hipMemcpy(h, g_dbg, sizeof(h), hipMemcpyDeviceToHost));
    for (int w : {0, 4}) {
      const unsigned long long* d = h + w * 8;
      const double kt = (double)d[7];
      printf("    wave %d per K-tile: load %.0f  bar1 %.0f  mfma %.0f  vmwait %.0f  bar2 %.0f  | total %.0f cyc/tile, clock %.0f MHz\n", w,
             d[0] / kt, d[1] / kt, d[2] / kt, d[3] / kt, d[4] / kt, d[5] / kt, (double)d[5] / (double)d[6] * 100.0);
      if (!(MODE & 128)) printf("      whole workgroup: prologue %llu  K loop %llu  epilogue (to stores retired) %llu cycles\n",
                                h[80 + w * 4 + 0], h[80 + w * 4 + 1], h[80 + w * 4 + 2]);
      if (MODE & 128) printf("      epilogue (store + reset) %.0f cyc per K-tile = %.0f per output tile (K-tiles per output tile %d)\n",
                             h[64 + w] / kt, (double)h[64 + w] / kt * (K / 64), K / 64);
    }
  }
  return tf;
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes = {{65536, 960, 320}, {65536, 2560, 320}, {16384, 1920, 640}, {16384, 640, 5760}};
  size_t maxX = 0, maxW = 0, maxC = 0;
  for (auto& s : shapes) { maxX = std::max(maxX, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
  std::vector<uint16_t> hX(maxX), hW(maxW), hC(maxC);
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  for (auto& v : hX) v = f2bf(rnd());
  for (auto& v : hW) v = f2bf(rnd());
  bf16 *dX, *dW, *dC;
  CK(hipMalloc(&dX, maxX * 2)); CK(hipMalloc(&dW, maxW * 2)); CK(hipMalloc(&dC, maxC * 2));
  CK(hipMemcpy(dX, hX.data(), maxX * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), maxW * 2, hipMemcpyHostToDevice));
  for (auto& s : shapes) {
    const int ntm = (s.M + 255) / 256;
    int gm = std::min(ntm, 8);
    CK(hipMemset(dC, 0, maxC * 2));
    auto check = [&]() {
      CK(hipMemcpy(hC.data(), dC, (size_t)s.M * s.N * 2, hipMemcpyDeviceToHost));
      double maxerr = 0; int bad = 0;
      uint32_t s2 = 777;
      for (int q = 0; q < 512; ++q) {
        s2 = s2 * 1664525u + 1013904223u; const int m = (s2 >> 4) % s.M;
        s2 = s2 * 1664525u + 1013904223u; const int n = (s2 >> 4) % s.N;
        double ref = 0;
        for (int k = 0; k < s.K; ++k) ref += (double)bf2f(hX[(size_t)m * s.K + k]) * bf2f(hW[(size_t)n * s.K + k]);
        const double got = bf2f(hC[(size_t)m * s.N + n]);
        const double err = fabs(got - ref) / (fabs(ref) + sqrt((double)s.K) * 0.05);
        if (err > maxerr) maxerr = err;
        if (err > 2e-2) ++bad;
      }
      printf("    check: max scaled err %.3e, bad %d / 512\n", maxerr, bad);
    };
    run<160, 2 + 64>("pp256x160 bal LAG", dX, dW, dC, s.M, s.N, s.K, gm, 20); check();
    run<160, 2 + 64 + 32>("pp256x160 bal LAG STAMP", dX, dW, dC, s.M, s.N, s.K, gm, 20);
  }
  return 0;
}
