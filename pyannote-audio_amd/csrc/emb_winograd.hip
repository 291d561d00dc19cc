// Winograd F(2x2, 3x3) for the stride-1 BasicBlock convolutions of the WeSpeaker ResNet on gfx950
// (reference: models/embedding/wespeaker/resnet.py:84-145; what MIOpen, the reference's backend, also
//  selects for fp32 3x3 convolutions).  2.25x fewer multiplies than the direct form at true fp32:
//
//     Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A         per 2x2 output tile, 4x4 input patch d
//
//   * U = G g G^T (BatchNorm scale folded into g) is precomputed on the host in float64 and stored as one
//     contiguous 32-KB LDS image per (32-cout slice, 16-cin stage) (weights.winograd_pack);
//   * a workgroup (4 waves, 2 workgroups per CU) owns TR x 16*TCG Winograd tiles x 32 output channels;
//     wave w owns 16 consecutive tiles of one tile row: lane (t = lane & 15, g = lane >> 4) transforms
//     the 4x4 patch of tile t for the channel quad g IN REGISTERS (V = B^T d B, 32 adds per channel) --
//     V never touches LDS -- and feeds it straight to v_mfma_f32_16x16x4_f32 as the A operand: for each
//     of the 16 transform points xi, D[tile][cout] += V_xi[tile][cin] * U_xi[cin][cout];
//   * the accumulators of the 16 points (16 x 2 cout groups x 4 VGPRs) stay in registers over the whole
//     cin loop.  U is the MFMA's A operand and V its B operand, so a lane ends up with FOUR CONSECUTIVE
//     output channels of one tile for all 16 points: the inverse transform A^T M A is lane-local float4
//     arithmetic (packed f32 adds; subtractions as v_pk_fma_f32 with an opaque -1) and the epilogue =
//     + BN shift (+ residual) (+ ReLU) moves 16 bytes per access (8 stores + 8 residual loads per lane
//     instead of 32 + 32), branch-free; on the 8 x 32-pixel tiles half of the residual travels through LDS
//     (LDS-DMA pieces issued in front of the tile's last MFMA run, which hides their latency: RPRE below);
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS without passing through VGPRs, the
//     hardware bounds check zero-fills the halo): the input patch, de-interleaved by column parity so
//     that the stride-2 tile walk reads consecutive LDS rows, and the U slab.  Rows are 64 B (16
//     channels) unpadded; the 16-B quad of row r sits at slot (g + 2*((r>>2)&1)) & 3, which makes every
//     ds_read_b128 of 16 consecutive rows bank-conflict free (brute-forced over all alignments).
//   * what bounds the kernel (tools/probes/pingpong_probe.py, profiles/r2_mfma_probe.txt): on gfx950 the
//     f32 MFMA executes at the vector rate ON the vector ALUs -- no VALU, LDS or VMEM instruction of either
//     wave of a SIMD issues under it -- so SIMD time is the SUM of 32 cycles per MFMA and the issue cycles
//     of everything else.  Hence: all per-lane address arithmetic is done once per kernel (1 VALU per DMA
//     piece), the first stage of a tile starts from the inline constant 0 instead of zeroing 128
//     accumulator registers, epilogue addresses are incremental, and two independent workgroups per CU
//     interleave at instruction granularity (a strict two-phase ping-pong of an 8-wave workgroup was
//     measured 6 % slower: tools/probes/emb_winograd_pingpong.hip.txt).
//   * where a stage's time goes (s_memtime stamps of an instrumented build, tools/wino_stamps.py, cycles per
//     wave and stage on the 20 x 250 x 128 layer): barrier 90 | DMA issue 3 100-4 700 | wait 450-600 |
//     barrier 200-270 | transform 700-1 240 | 128 MFMAs 4 190 | epilogue 1 000 (amortised) = 11 100, against
//     2 x 4 096 for the two waves of a SIMD.  The DMA phase is long because its 15 instructions only issue in
//     the gaps of the OTHER workgroup's MFMA stream (same for the epilogue's and the transform's vector
//     instructions); a wave's own DMA spread through its own MFMA run costs 5-25 cycles per instruction
//     (tools/probes/interleave_probe.py) but needs both LDS images double-buffered = one workgroup per CU.
//     Raising the wave priority outside the MFMA run (s_setprio) does not change the picture.  That
//     one-workgroup-per-CU software-pipelined form was built (tools/probes/emb_winograd_sp.hip.txt: next
//     stage's DMA and the residual loads issued from inside the MFMA run, one barrier per stage, results
//     identical) and measured 10-17 % SLOWER (6 350 instead of 5 430 cycles per stage and wave on the
//     128-channel layer): with nothing else on the SIMD its transform, descriptor arithmetic, barrier and
//     epilogue are all exposed.  Two workgroups per CU stay.
// Numerics: fp32 throughout; the transforms only add/subtract and the 1/2 factors of G are applied in
// float64 on the host; error ~3x the direct form's (tests: |err| <= 1e-4 max|ref| per conv, end to end).
#include <stdlib.h>

#include "common.h"
#ifndef PA_WINO32_PATCH_FIRST   // k_conv3x3_wino32: the next tile's patch DMA in front of the epilogue (0: behind it, round 3)
#define PA_WINO32_PATCH_FIRST 1
#endif
#ifndef PA_WINO32_WAIT_STORES   // 1: the step barrier of k_conv3x3_wino32 also waits for the epilogue's stores (A/B aid)
#define PA_WINO32_WAIT_STORES 0
#endif
#ifndef PA_WINO_STORE_AUX   // cache-policy bits of the output stores (2 = nt): A/B aid, see profiles/r5_xcd_ranges.txt
#define PA_WINO_STORE_AUX 0
#endif

#ifdef PA_WINO_NOSCHED   // development A/B switch (never defined in the product build)
#define WINO_SCHED_BARRIER()
#else
#define WINO_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

#ifndef PA_WINO_STAMP
#define PA_WINO_STAMP 0
#endif
// (nt / "streaming" cache-policy bits on the patch DMA, the residual loads or the stores: measured neutral to
//  10-25 % slower, profiles/r3_wino_cache_policy.txt -- every access keeps the default policy)
#ifndef PA_WINO_RPIN
#define PA_WINO_RPIN 1
#endif
#ifndef PA_WINO_REFRESH   // 128-channel residual kernel with pinned residual loads (-DPA_WINO_REFRESH=0: A/B)
#define PA_WINO_REFRESH 1
#endif
#ifndef PA_WINO_RTOUCH
#define PA_WINO_RTOUCH 1
#endif
#ifndef PA_WINO_RPRE   // residual prefetch through LDS in front of a tile's last MFMA run (-DPA_WINO_RPRE=0: A/B)
#define PA_WINO_RPRE 1
#endif
namespace pa {

#if PA_WINO_STAMP
// development instrumentation (never in the product build): s_memtime stamps of the phases of the first
// 64 stages of the first 16 workgroups, per wave; read back with pa_wino_read_stamps
constexpr int STAMP_WG = 16, STAMP_IT = 64, STAMP_PH = 8;
__device__ unsigned long long g_wino_stamps[STAMP_WG * 4 * STAMP_IT * STAMP_PH];
#define WINO_STAMP(p) st_[p] = __builtin_amdgcn_s_memtime()
#define WINO_STAMP_FLUSH()                                                                    \
  do {                                                                                        \
    if (blockIdx.x < STAMP_WG && st_iter < STAMP_IT && lane == 0) {                            \
      _Pragma("unroll") for (int p_ = 0; p_ < STAMP_PH; ++p_)                                  \
          g_wino_stamps[((blockIdx.x * 4 + slw) * STAMP_IT + st_iter) * STAMP_PH + p_] = st_[p_]; \
    }                                                                                         \
    ++st_iter;                                                                                \
  } while (0)
#else
#define WINO_STAMP(p)
#define WINO_STAMP_FLUSH()
#endif

}  // namespace pa
#include "emb_winograd_geom.h"
namespace pa {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Workgroup barrier WITHOUT the release/acquire fence of __syncthreads(): the fence drains vmcnt, i.e. it
// would wait for the epilogue's global stores and for the patch that is being prefetched.  What has to
// be ordered here is LDS only: a wave's ds_reads have returned before it gets here (their results were
// consumed), and the LDS-DMA writes are awaited explicitly (s_waitcnt vmcnt) before the barrier that
// publishes them.  The empty asm statements keep the compiler from moving memory operations across.
__device__ __forceinline__ void wino_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int TR, int TCG>
__device__ __forceinline__ void wino_issue_patch(const float* __restrict__ X, int H, int W, int CIN,
                                                 const WinoTile& q, int c0, float* patch, const int* prel,
                                                 int slw, int x0_last) {
  using G = WinoGeom<TR, TCG>;
  const long img = (long)H * W * CIN;                              // floats per image
  const long org = ((long)(q.y0 - 1) * W + (q.x0 - 1)) * CIN + c0;  // patch origin within the image
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(X + (long)q.b * img + org), 0, (int)((img - org) * 4), 0x00020000);
  const int keep = wino_patch_keep(q, x0_last);
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    const int k = slw + HW * i;
    if (k >= G::PINSTR) break;  // wave-uniform
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrd, (lds_ptr_t)(patch + 256 * k), 16, prel[i] & keep, 0, 0, 0);
  }
}

// U slab: the host stores every (32-cout slice, 16-cin stage) slab as one contiguous 32-KB image of the
// LDS layout (weights.winograd_pack), so a piece is a linear 1-KB burst: lane l copies 16 bytes at
// slab + 1024 k + 16 l; 8 pieces per wave.
__device__ __forceinline__ void wino_issue_u(const float* __restrict__ U, int CIN, int COUT, int n0, int c0,
                                             float* uslab, int lane, int slw) {
  const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(U), 0, 16 * COUT * CIN * 4, 0x00020000);
  const int slab = (n0 / W_BN) * (CIN / WCB) + c0 / WCB;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = slw + HW * i;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(usrd, (lds_ptr_t)(uslab + 256 * k), 16, lane * 16,
                                             (slab * 8192 + 256 * k) * 4, 0, 0);
  }
}

// a - b on a float4 as two v_pk_fma_f32 (b * (-1) + a, exact).  The compiler scalarises a vector
// subtraction into four v_sub_f32 (there is no packed subtract) and folds a literal -1 back into one, so the
// -1 comes from an opaque scalar move; every vector instruction here is paid in matrix time.
__device__ __forceinline__ float wino_minus_one() {
  float m;
  asm("s_mov_b32 %0, 0xbf800000" : "=s"(m));
  return m;
}
__device__ __forceinline__ f32x4 vsub(const f32x4 a, const f32x4 b, const float m1) {
  const f32x4 m = {m1, m1, m1, m1};
  return __builtin_elementwise_fma(b, m, a);
}

// input transform V = B^T d B of tile (wr, 16*wc + t) for channels 4g..4g+3, in registers
// PIPE: patch row i+1 is read while row i is combined (two register sets of 4 x f32x4) -- hipcc, short of
// registers, otherwise waits for every group of four reads before it issues the next one.  Measured per launch
// (profiles/r3_wino_transform_pipelined.txt): -0.5 ... -3.3 % on the launches without a residual input, +4 ... 6 %
// (spills) on those with one, so the kernel asks for it only when HAS_R is false.
template <int TR, int TCG, bool PIPE = false>
__device__ __forceinline__ void wino_transform(const float* patch, const int (&pbase)[8], f32x4 (&v)[4][4],
                                               float m1) {
  using G = WinoGeom<TR, TCG>;
  const char* base = reinterpret_cast<const char*>(patch);
  if (PIPE) {
  f32x4 d[2][4];
  auto rd = [&](int i, f32x4 (&o)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int K = (2 * i + (j & 1)) * G::PWH + (j >> 1);
      o[j] = *reinterpret_cast<const f32x4*>(base + pbase[K & 7] + (K & ~7) * (WCB * 4));
    }
  };
  rd(0, d[0]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i + 1 < 4) rd(i + 1, d[(i + 1) & 1]);
    WINO_SCHED_BARRIER();
    const f32x4(&c)[4] = d[i & 1];
    v[i][0] = vsub(c[0], c[2], m1);
    v[i][1] = c[1] + c[2];
    v[i][2] = vsub(c[2], c[1], m1);
    v[i][3] = vsub(c[1], c[3], m1);
    WINO_SCHED_BARRIER();
  }
  } else {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int K = (2 * i + (j & 1)) * G::PWH + (j >> 1);
      d[j] = *reinterpret_cast<const f32x4*>(base + pbase[K & 7] + (K & ~7) * (WCB * 4));
    }
    v[i][0] = vsub(d[0], d[2], m1);
    v[i][1] = d[1] + d[2];
    v[i][2] = vsub(d[2], d[1], m1);
    v[i][3] = vsub(d[1], d[3], m1);
  }
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    const f32x4 r0 = v[0][bb], r1 = v[1][bb], r2 = v[2][bb], r3 = v[3][bb];
    v[0][bb] = vsub(r0, r2, m1);
    v[1][bb] = r1 + r2;
    v[2][bb] = vsub(r2, r1, m1);
    v[3][bb] = vsub(r1, r3, m1);
  }
}

// 16 transform points x 2 cout groups x 4 k-steps; B fragments one point ahead of their MFMAs (a second
// point of look-ahead does not fit the 256-VGPR budget, and what the reads cost is issue time, not latency).
// FIRST: the first stage of a tile starts the accumulators from the inline constant 0 (no v_mov run).
struct WinoNoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// `hook(xi)` runs after the MFMAs of point xi: the software-pipelined kernel issues the NEXT stage's DMA
// pieces (and the residual loads of the epilogue) from there -- a wave's own memory instructions cost it
// 5-25 cycles each inside its MFMA run (tools/probes/interleave_probe.py).
template <bool FIRST, typename Hook = WinoNoHook>
__device__ __forceinline__ void wino_mfma(const float* uslab, const f32x4 (&v)[4][4], f32x4 (&acc)[16][2],
                                          int t, int g, const Hook& hook = Hook()) {
  constexpr int PF = 1;
  f32x4 bf[PF + 1][2];
  // row = 32 xi + 16 cg + t: its swizzle bit ((row >> 2) & 1) = (t >> 2) & 1 does not depend on xi / cg,
  // so every read is (lane base) + compile-time offset -> folded into the ds_read immediate
  const float* ub = uslab + t * WCB + 4 * wslot(t, g);
  auto load = [&](int xi) {
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
      bf[xi % (PF + 1)][cg] = *reinterpret_cast<const f32x4*>(ub + (xi * W_BN + 16 * cg) * WCB);
  };
#pragma unroll
  for (int xi = 0; xi < PF; ++xi) load(xi);
  // sched_barrier(0): nothing moves across.  Without them the scheduler (minimising register pressure)
  // sinks each ds_read pair right in front of its first MFMA and the LDS latency is exposed 16 times per
  // stage (measured: 46 instead of 32 cycles per MFMA).
  WINO_SCHED_BARRIER();
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    if (xi + PF < 16) load(xi + PF);
    WINO_SCHED_BARRIER();
    const f32x4 av = v[xi >> 2][xi & 3];
    // alternate the two accumulators: a 16x16x4 MFMA issues every 32 cycles but its result is ready
    // after 40, so back-to-back MFMAs on ONE accumulator would stall 8 cycles each
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // U as the A operand (rows = output channels), V as B (columns = tiles): D[cout 4g + r][tile t],
      // i.e. a lane ends up with FOUR CONSECUTIVE output channels of ONE tile -> 16-byte epilogue accesses
      acc[xi][0] = MFMA16(bf[xi % (PF + 1)][0][ks], av[ks], (FIRST && ks == 0) ? zero : acc[xi][0]);
      acc[xi][1] = MFMA16(bf[xi % (PF + 1)][1][ks], av[ks], (FIRST && ks == 0) ? zero : acc[xi][1]);
    }
    WINO_SCHED_BARRIER();
    hook(xi);
    WINO_SCHED_BARRIER();
  }
}

// inverse transform + epilogue.
// acc[xi][cg] (f32x4): tile (row wr, column 16wc + t), output channels n0 + 16cg + 4g + {0..3}: the four
// outputs (y0 + 2wr + p, x0 + 2(16wc + t) + qq) of the tile are 16-byte vectors of consecutive channels in
// the NHWC map -> 8 buffer_store_dwordx4 (+ 8 residual buffer_load_dwordx4, all issued before the inverse
// transform so that their latency hides under its arithmetic) per lane instead of 32 + 32 dword accesses,
// and the inverse transform A^T M A runs on float4 (packed f32 adds).  Rows below the image are past
// num_records by themselves; the column bound needs one compare per output column.
// byte offsets of the tile's four output pixels for this lane (channels n0 + 4g .., cg adds 64 bytes)
__device__ __forceinline__ void wino_out_offsets(int (&off)[4], const WinoTile& q, int W, int COUT, int t, int g,
                                                 int wr, int wc) {
  constexpr int OOB = (int)0x80000000;
  const int xl = q.x0 + 2 * (16 * wc + t);
  const bool in0 = q.valid && xl < W, in1 = q.valid && xl + 1 < W;
  const int obase = (((q.y0 + 2 * wr) * W + xl) * COUT + q.n0 + 4 * g) * 4;
  const int srow = W * COUT * 4, spix = COUT * 4;
  off[0] = in0 ? obase : OOB;
  off[1] = in1 ? obase + spix : OOB;
  off[2] = in0 ? obase + srow : OOB;
  off[3] = in1 ? obase + srow + spix : OOB;
}

// PRE: the residual values were loaded by the caller (inside the last MFMA run) into `rv`
// PRE0: the first channel group's four residual vectors were sent to LDS (`rbuf`, 1 KB per wave and vector, lane l
// at 16 l) by LDS-DMA pieces issued in front of the tile's LAST MFMA run, so that their HBM latency hides under
// that run without holding registers (the kernel has none to spare: a register version spilled 20-30 VGPRs);
// the second group's loads are issued here and hide under the first group's inverse transform.
// 16-byte stores of one epilogue per lane: 2 channel groups x the 4 outputs of the F(2x2) tile -- the NEWEST vector
// memory operations of whoever calls it (k_conv3x3_wino32 counts on that: WINO_EPILOGUE_STORES_LIT in its step wait)
constexpr int WINO_EPILOGUE_STORES = 2 * 4;
#define WINO_EPILOGUE_STORES_LIT 8
static_assert(WINO_EPILOGUE_STORES == WINO_EPILOGUE_STORES_LIT, "the literal of the s_waitcnt string");
#define WINO_STR2(x) #x
#define WINO_STR(x) WINO_STR2(x)
template <bool HAS_R, bool PRE = false, bool PRE0 = false, bool PIN = false>
__device__ __forceinline__ void wino_epilogue(const f32x4 (&acc)[16][2], const WinoTile& q, int H, int W,
                                              int COUT, const float* __restrict__ shift,
                                              const float* __restrict__ R, float* __restrict__ Y,
                                              int relu, int t, int g, int wr, int wc, float m1,
                                              const int* off_pre = nullptr, f32x4 (*rv_pre)[4] = nullptr,
                                              const float* rbuf = nullptr) {
  const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(
      Y + (long)q.b * H * W * COUT, 0, H * W * COUT * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HAS_R ? R + (long)q.b * H * W * COUT : Y), 0, H * W * COUT * 4, 0x00020000);
  int off[4];
  if (PRE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) off[e] = off_pre[e];
  } else {
    wino_out_offsets(off, q, W, COUT, t, g, wr, wc);
  }
  f32x4 rv[2][4];
#pragma unroll
  for (int cgi = 0; cgi < 2; ++cgi)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cg = (PRE0 && HAS_R) ? 1 - cgi : cgi;   // PRE0: the global loads of group 1 go out first
      if (PRE && HAS_R) rv[cg][e] = rv_pre[cg][e];
      else if (PRE0 && HAS_R && cg == 0) {
        if (e == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the 4 loads of cg = 1 are newer
        rv[0][e] = *reinterpret_cast<const f32x4*>(rbuf + 256 * e);
      } else
        rv[cg][e] = HAS_R ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrd, off[e] + 64 * cg, 0, 0))
                          : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // all residual loads leave BEFORE the inverse transform: left alone, the scheduler sinks some of them next to
  // their first use (seen in the 4 x 64-pixel instantiation: two loads each followed by a full vmcnt(0) wait, i.e.
  // two exposed HBM latencies per tile)
  if (HAS_R && PIN) __builtin_amdgcn_sched_barrier(0);
  const float lo = relu ? 0.f : -__builtin_inff();
  const f32x4 lo4 = {lo, lo, lo, lo};
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + q.n0 + 16 * cg + 4 * g);
    f32x4 s[4], dd[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      s[a] = acc[4 * a + 0][cg] + acc[4 * a + 1][cg] + acc[4 * a + 2][cg];
      dd[a] = vsub(vsub(acc[4 * a + 1][cg], acc[4 * a + 2][cg], m1), acc[4 * a + 3][cg], m1);
    }
    f32x4 o4[4];
    o4[0] = s[0] + s[1] + s[2];
    o4[1] = dd[0] + dd[1] + dd[2];
    o4[2] = vsub(vsub(s[1], s[2], m1), s[3], m1);
    o4[3] = vsub(vsub(dd[1], dd[2], m1), dd[3], m1);
    static_assert(WINO_EPILOGUE_STORES == 2 * 4, "2 channel groups x 4 stores below");
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 vv = __builtin_elementwise_max(o4[e] + sh + rv[cg][e], lo4);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), ysrd, off[e] + 64 * cg, 0, PA_WINO_STORE_AUX);
    }
  }
}

// One LDS buffer per workgroup, two workgroups per CU: while one waits for its stage to land or runs its
// input transform, the other one issues MFMAs.
template <int TR, int TCG, bool HAS_R>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wino(
    const float* __restrict__ X, int H, int W, int CIN, const float* __restrict__ U,
    const float* __restrict__ shift, const float* __restrict__ R, float* __restrict__ Y, int COUT,
    int relu, int tiles_w, int tiles_hw, int n_tiles, int total_tiles, int num_pb, int y_first, int xranges,
    int* __restrict__ counters) {
  using G = WinoGeom<TR, TCG>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int slw = __builtin_amdgcn_readfirstlane(wv);
  const int wr = slw / TCG, wc = slw % TCG;  // tile row / 16-tile column group of this wave
  float* patch = smem;
  float* uslab = smem + G::PATCH;

  // residual staging (RPRE): 4 x 1 KB per wave behind the mailbox, for the 8 x 32-pixel tiles of the 32- and
  // 64-channel layers (few stages per tile: the residual's latency was 14 % / 8 % of a tile).  Measured per
  // launch (B = 512, profiles/r3_wino_residual_prefetch.txt): 80x998x32 4.91 -> 4.44 ms, 40x499x64 3.77 -> 3.55;
  // on the 4 x 64-pixel tiles (128 channels) it LOSES 1 % and the 2 x 128-pixel tiles have no LDS left for it.
  constexpr bool RPRE = PA_WINO_RPRE && HAS_R && TCG == 1;
  float* rbuf = smem + G::LDS_FLOATS + 4 + 1024 * slw;
  // the 2 x 128-pixel tiles (256 channels) only TOUCH their residual lines ahead of the tile's last MFMA run (one
  // 4-byte DMA piece per wave into a 256-B scratch row that nobody reads): the epilogue's loads then hit L2.
  // Measured (B = 512, profiles/r3_wino_residual_touch.txt): 10x125x256 3.157 -> 3.099 ms; the 4 x 64-pixel tiles
  // (128 channels) LOSE 2 % with it (3.18 -> 3.24 ms), like they lose with the LDS staging above.
  constexpr bool RTOUCH = PA_WINO_RTOUCH && HAS_R && TCG == 4;
  // tiles are CLAIMED, not statically strided (common.h: TileQueue): thread 0 claims the next tile while
  // the current one is processed and publishes it through LDS at the tile boundary
  int* s_next = reinterpret_cast<int*>(smem + G::LDS_FLOATS);
  const TileQueue tq{counters, (int)(blockIdx.x & 7), total_tiles >> 3};
  if (tid == 0) *s_next = tq_resolve(tq, tq_claim_own(tq));
  __syncthreads();
  int q = *s_next;
  int ahead = 0;
  const int x0_last = (tiles_w - 1) * 32 * TCG;
  // The 4 x 64-pixel residual instantiation (128-channel layers) only: the compiler sank two of its eight residual
  // loads next to their first use, each behind a full `s_waitcnt vmcnt(0)` -- three exposed HBM latencies per tile
  // (ISA of round 3's measured build; its twin without residual is 10 % faster per launch).  There the loads are
  // pinned in front of the inverse transform (wino_epilogue<.., PIN>) and the lane constants of the patch DMA /
  // transform are rebuilt per tile so that they are dead across the epilogue: 8 loads in flight, ONE wait after the
  // first channel group's transform, 243 VGPRs and no spill (was 256 with 3 spills).  Measured in round 4 (B = 512,
  // 20x250x128 + residual, two runs on one box): 3.169 -> 3.136 ms and 3.177 -> 3.127 ms (-1.3 %), results identical
  // (tests/test_emb_gpu.py); tools/build_variants.py "norefresh" builds the old form.  The other instantiations do
  // not change.
  constexpr bool REFRESH = PA_WINO_REFRESH && HAS_R && TR == 2 && TCG == 2;
  int prel[G::NPP];
  if (!REFRESH) wino_patch_lanes<TR, TCG>(prel, W, CIN, lane, slw, x0_last);
  f32x4 acc[16][2];
  int pbase[8];
  if (!REFRESH) wino_patch_bases<TR, TCG>(pbase, t, g, wr, wc);
  const float m1 = wino_minus_one();
#if PA_WINO_STAMP
  unsigned long long st_[STAMP_PH] = {0, 0, 0, 0, 0, 0, 0, 0};
  int st_iter = 0;
#endif
  {
    while (q >= 0) {
      if (REFRESH) {
        // the lane constants of the patch DMA / transform are rebuilt per tile (an opaque copy of the lane number
        // keeps the compiler from hoisting them back): they are then dead across the epilogue, which needs the
        // registers to keep all eight residual vectors in flight
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        wino_patch_lanes<TR, TCG>(prel, W, CIN, lane_v, slw, x0_last);
        wino_patch_bases<TR, TCG>(pbase, lane_v & 15, lane_v >> 4, wr, wc);
      }
      const WinoTile cur = wino_decode(q, tiles_w, tiles_hw, n_tiles, 2 * TR, 32 * TCG, num_pb, y_first, xranges);
      if (tid == 0) ahead = tq_claim_own(tq);
      for (int c0 = 0; c0 < CIN; c0 += WCB) {
        WINO_STAMP(0);
        wino_barrier();  // every wave is done reading the previous stage
        WINO_STAMP(1);
        wino_issue_patch<TR, TCG>(X, H, W, CIN, cur, c0, patch, prel, slw, x0_last);
        wino_issue_u(U, CIN, COUT, cur.n0, c0, uslab, lane, slw);
        WINO_STAMP(2);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        WINO_STAMP(3);
        wino_barrier();
        WINO_STAMP(4);
        f32x4 v[4][4];
        wino_transform<TR, TCG, !HAS_R>(patch, pbase, v, m1);
#if PA_WINO_STAMP
        asm volatile("s_nop 0" ::"v"(v[3][3]), "v"(v[0][0]));   // the transform is complete here
#endif
        WINO_STAMP(5);
        if (RPRE && c0 + WCB >= CIN) {
          // last stage of the tile: the residual vectors of the first channel group leave for LDS now, ahead
          // of the MFMA run that hides their latency (wino_epilogue<.., PRE0>)
          int off_pre[4];
          wino_out_offsets(off_pre, cur, W, COUT, t, g, wr, wc);
          const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(R + (long)cur.b * H * W * COUT), 0, H * W * COUT * 4, 0x00020000);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrd, (lds_ptr_t)(rbuf + 256 * e), 16, off_pre[e], 0, 0, 0);
        }
        if (RTOUCH && c0 + WCB >= CIN) {
          // one pixel (= one 128-B line of this workgroup's 32 output channels) per thread; its own few
          // integer operations, so that nothing of the epilogue's offsets is kept alive across the MFMA run
          // (wave slw covers 64 consecutive pixels of tile row 64 slw / (32 TCG): scalar base + lane * COUT * 4)
          const int yy = cur.y0 + (64 * slw) / (32 * TCG), xb = cur.x0 + (64 * slw) % (32 * TCG);
          const int sbase = ((yy * W + xb) * COUT + cur.n0) * 4;
          const int off = (yy < H && xb + lane < W) ? sbase + lane * COUT * 4 : H * W * COUT * 4;
          const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<float*>(R + (long)cur.b * H * W * COUT), 0, H * W * COUT * 4, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrd, (lds_ptr_t)(smem + G::LDS_FLOATS + 4 + 64 * slw), 4, off, 0,
                                                   0, 0);
        }
        if (c0 == 0) wino_mfma<true>(uslab, v, acc, t, g);
        else wino_mfma<false>(uslab, v, acc, t, g);
        WINO_STAMP(6);
        if (c0 + WCB < CIN) {
          WINO_STAMP(7);
          WINO_STAMP_FLUSH();
        }
      }
      // (Issuing the NEXT tile's first stage in front of this epilogue, so that the epilogue hides its flight
      // time, was measured 3-8 % SLOWER on every layer shape: profiles/r3_wino_next_tile_prefetch.txt -- the
      // extra DMA issue lands in the phase where the other workgroup's MFMA stream owns the SIMD.)
      wino_epilogue<HAS_R, false, RPRE, REFRESH && PA_WINO_RPIN>(acc, cur, H, W, COUT, shift, R, Y, relu, t, g, wr, wc,
                                                                 m1, nullptr, nullptr, rbuf + 4 * lane);
      WINO_STAMP(7);
      WINO_STAMP_FLUSH();
      if (tid == 0) *s_next = tq_resolve(tq, ahead);
      __syncthreads();
      q = *s_next;
    }
  }
  if (tid == 0) tq_done(tq, gridDim.x);
}

int xcd_ranges_wanted(bool by_default);   // emb_winograd4.hip

template <int TR, int TCG, bool HAS_R>
static int launch_wino_r(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                         const float* R, float* Y, int COUT, int relu, int y_first, hipStream_t st) {
  using G = WinoGeom<TR, TCG>;
  const int tiles_w = cdiv(W, 32 * TCG), tiles_h = cdiv(H - y_first, 2 * TR);
  // + the claimed-tile mailbox (+ 16 KB of residual staging where it is used: k_conv3x3_wino RPRE)
  const size_t lds = (size_t)G::LDS_FLOATS * sizeof(float) + 16 +
                     ((PA_WINO_RPRE && HAS_R && TCG == 1) ? 4 * 4096 : (PA_WINO_RTOUCH && HAS_R && TCG == 4 ? 4 * 256 : 0));
  auto kernel = k_conv3x3_wino<TR, TCG, HAS_R>;
  // per-device launch state (the attribute and the CU count belong to a device, not to the process)
  constexpr int MAXDEV = 16;
  static int resident_of[MAXDEV] = {0}, per_cu_of[MAXDEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAXDEV) dev = 0;
  if (!resident_of[dev]) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int cus = 256, per_cu = 2;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) !=
            hipSuccess || per_cu < 1)
      per_cu = 2;
    resident_of[dev] = cus * per_cu;
    per_cu_of[dev] = per_cu;
  }
  const int tiles_hw = tiles_w * tiles_h, n_tiles = COUT / W_BN;
  const long num_pb = (long)tiles_hw * B;                    // (pixel tile, image) pairs
  const long total = ((num_pb + 7) / 8) * 8 * n_tiles;       // padded to whole XCD stripes
  // every resident workgroup is launched whatever else runs on the chip: tiles are claimed at run time, so
  // a workgroup that is placed late (a foreign kernel holds its CU) costs nothing but its own absence
  const int resident = resident_of[dev] & ~7;
  const int grid = (int)(total < resident ? total : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3_wino: cannot allocate the tile counters");
    return 2;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, st, X, H, W, CIN, U, shift, R, Y, COUT, relu, tiles_w,
                     tiles_hw, n_tiles, (int)total, (int)num_pb, y_first, xcd_ranges_wanted(true), counters);
  return 0;
}

// =============================================================================================
// CIN = COUT = 32 (layer 1 of the ResNet: 6 of the 29 Winograd launches, a quarter of their time).  With only two
// channel stages per tile the generic kernel spends as long in DMA issue / waits / barriers / epilogue as in its
// MFMA runs (0.49-0.52 of peak against 0.69-0.74 on the deeper layers).  Here
//   * both U slabs (64 KB) are loaded ONCE per workgroup and stay in LDS: no weight DMA per tile at all;
//   * a workgroup has 8 waves = two HALVES of 4 waves, each with its own 8 x 32-pixel tile and a patch buffer
//     for BOTH channel stages (2 x 22 KB), staged by one DMA phase per tile;
//   * the halves alternate: while one runs its two transform + MFMA stages back to back (no DMA, no barrier
//     inside), the other does the epilogue of its previous tile -- the residual loads' and the stores' latency
//     hide under the partner's MFMA run by construction -- and stages the patch of its next tile.
//     Two workgroup barriers per pair of tiles.
//   * tiles are claimed per half (TileQueue); the claim is issued in a PREPARE step, resolved in the next
//     COMPUTE step and published by that step's closing barrier.
// One workgroup per CU (152 KB of LDS), two waves per SIMD as before.  Same arithmetic in the same order as the generic
// kernel: bit-identical outputs.  Measured (B = 512, profiles/r3_wino32_ab.txt): 80x998x32 3.83 -> 3.62 ms without
// residual (0.59 of peak), 4.35 -> 4.13 ms with (0.52) -- less than the schedule promises, because the two halves
// share every SIMD and f32 MFMAs issue at the vector rate: SIMD time is the SUM of both halves' issue cycles.
// PA_WINO32=0 (environment) selects the generic kernel for an A/B.
// =============================================================================================
template <bool HAS_R>
__global__ __launch_bounds__(512) void k_conv3x3_wino32(
    const float* __restrict__ X, int H, int W, const float* __restrict__ U, const float* __restrict__ shift,
    const float* __restrict__ R, float* __restrict__ Y, int relu, int tiles_w, int tiles_hw, int total_tiles,
    int num_pb, int xranges, int* __restrict__ counters) {
  using G = WinoGeom<4, 1>;
  constexpr int CIN = 32, COUT = 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = lane & 15, g = lane >> 4;
  const int half = __builtin_amdgcn_readfirstlane(wv >> 2);
  const int slw = __builtin_amdgcn_readfirstlane(wv & 3);
  const int wr = slw, wc = 0;
  float* uslab = smem;                                           // [2 stages][USLAB]
  float* patch = smem + 2 * G::USLAB + half * 2 * G::PATCH;      // [2 stages][PATCH] of this half
  int* mail = reinterpret_cast<int*>(smem + 2 * G::USLAB + 4 * G::PATCH);   // [0..1] next tile, [2..3] alive
  const bool leader = (tid & 255) == 0;

  // the two U slabs: 64 pieces of 1 KB, 8 per wave, contiguous in the packed weight image
  {
    const __amdgpu_buffer_rsrc_t usrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(U), 0, 16 * COUT * CIN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = wv + 8 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(usrd, (lds_ptr_t)(uslab + 256 * k), 16, lane * 16, 256 * k * 4, 0, 0);
    }
  }
  const TileQueue tq{counters, (int)(blockIdx.x & 7), total_tiles >> 3};
  if (leader) {
    mail[half] = tq_resolve(tq, tq_claim_own(tq));
    mail[2 + half] = 1;
  }
  const int x0_last = (tiles_w - 1) * 32;
  int prel[G::NPP];
  wino_patch_lanes<4, 1>(prel, W, CIN, lane, slw, x0_last);
  int pbase[8];
  wino_patch_bases<4, 1>(pbase, t, g, wr, wc);
  const float m1 = wino_minus_one();
  f32x4 acc[16][2];
  __syncthreads();
  int q = mail[half];          // tile being staged / computed by this half (-1: none left)
  int done_q = -1;             // tile whose accumulators are waiting for their epilogue
  int ahead = 0;
  WinoTile cur = wino_decode(q < 0 ? 0 : q, tiles_w, tiles_hw, 1, 8, 32, num_pb, 0, xranges), fin = cur;
  if (q >= 0) {
    wino_issue_patch<4, 1>(X, H, W, CIN, cur, 0, patch, prel, slw, x0_last);
    wino_issue_patch<4, 1>(X, H, W, CIN, cur, WCB, patch + G::PATCH, prel, slw, x0_last);
    if (leader) ahead = tq_claim_own(tq);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  wino_barrier();
  // step s: half h COMPUTES when (s - h) is even and >= 0, otherwise it PREPARES
  for (int step = 0;; ++step) {
    const bool compute = step >= half && ((step - half) & 1) == 0;
    bool stored = false;        // this step ended with the stores of an epilogue (wave-uniform)
    if (compute) {
      if (q >= 0) {
        if (leader) mail[half] = tq_resolve(tq, ahead);   // the claim was issued a step ago: no round trip here
#pragma unroll
        for (int stage = 0; stage < 2; ++stage) {
          f32x4 v[4][4];
          wino_transform<4, 1, !HAS_R>(patch + stage * G::PATCH, pbase, v, m1);
          if (stage == 0) wino_mfma<true>(uslab, v, acc, t, g);
          else wino_mfma<false>(uslab + G::USLAB, v, acc, t, g);
        }
        done_q = q;
        fin = cur;
      } else if (leader) {
        mail[2 + half] = 0;     // nothing left for this half (its last epilogue ran in the previous step)
      }
    } else if (step >= half) {
      // PREPARE: epilogue of the tile computed in the previous step, then the patch of the next one
      const int qn = done_q >= 0 ? mail[half] : -1;   // (published by the barrier that closed the COMPUTE step)
#if PA_WINO32_PATCH_FIRST
      // the next tile's patch goes out FIRST (its buffers are free: this half's COMPUTE step has read them), so that its
      // flight overlaps the epilogue instead of following it: the layer is at 60-65 % of both of its bounds for lack of
      // loads in flight (profiles/r5_hbm_bw.txt)
      q = qn;
      if (q >= 0) {
        cur = wino_decode(q, tiles_w, tiles_hw, 1, 8, 32, num_pb, 0, xranges);
        wino_issue_patch<4, 1>(X, H, W, CIN, cur, 0, patch, prel, slw, x0_last);
        wino_issue_patch<4, 1>(X, H, W, CIN, cur, WCB, patch + G::PATCH, prel, slw, x0_last);
        if (leader) ahead = tq_claim_own(tq);
      }
      if (done_q >= 0) {
        wino_epilogue<HAS_R>(acc, fin, H, W, COUT, shift, R, Y, relu, t, g, wr, wc, m1);
        done_q = -1;
        stored = true;
      }
#else
      if (done_q >= 0) {
        wino_epilogue<HAS_R>(acc, fin, H, W, COUT, shift, R, Y, relu, t, g, wr, wc, m1);
        done_q = -1;
      }
      q = qn;
      if (q >= 0) {
        cur = wino_decode(q, tiles_w, tiles_hw, 1, 8, 32, num_pb, 0, xranges);
        wino_issue_patch<4, 1>(X, H, W, CIN, cur, 0, patch, prel, slw, x0_last);
        wino_issue_patch<4, 1>(X, H, W, CIN, cur, WCB, patch + G::PATCH, prel, slw, x0_last);
        if (leader) ahead = tq_claim_own(tq);
      }
#endif
    }
    // what the barrier needs is this half's patch in LDS.  The epilogue's 8 stores per lane are the NEWEST vector memory
    // operations of a PREPARE step (the patch pieces and the residual loads are older) and complete in order: leaving
    // them in flight across the barrier takes their acknowledgement latency off every step's critical path.
    if (PA_WINO32_PATCH_FIRST && !PA_WINO32_WAIT_STORES && stored)
      asm volatile("s_waitcnt vmcnt(" WINO_STR(WINO_EPILOGUE_STORES_LIT) ") lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wino_barrier();
    if (mail[2] == 0 && mail[3] == 0) break;
  }
  if (tid == 0) tq_done(tq, gridDim.x);
}

template <bool HAS_R>
static int launch_wino32(const float* X, int B, int H, int W, const float* U, const float* shift, const float* R,
                         float* Y, int relu, hipStream_t st) {
  using G = WinoGeom<4, 1>;
  const int tiles_w = cdiv(W, 32), tiles_h = cdiv(H, 8);
  const size_t lds = (size_t)(2 * G::USLAB + 4 * G::PATCH) * sizeof(float) + 32;
  auto kernel = k_conv3x3_wino32<HAS_R>;
  constexpr int MAXDEV = 16;
  static int cus_of[MAXDEV] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= MAXDEV) dev = 0;
  if (!cus_of[dev]) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus_of[dev] = cus;
  }
  const int tiles_hw = tiles_w * tiles_h;
  const long num_pb = (long)tiles_hw * B;
  const long total = ((num_pb + 7) / 8) * 8;
  const int resident = cus_of[dev] & ~7;            // one workgroup (two halves) per CU
  const long want = (total + 1) / 2;
  const int grid = (int)(want < resident ? ((want + 7) & ~7L) : resident);
  int* counters = tile_counters();
  if (counters == nullptr) {
    set_error("pa_conv3x3_wino: cannot allocate the tile counters");
    return 2;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, st, X, H, W, U, shift, R, Y, relu, tiles_w, tiles_hw,
                     (int)total, (int)num_pb, xcd_ranges_wanted(true), counters);
  return 0;
}

template <int TR, int TCG>
static int launch_wino(const float* X, int B, int H, int W, int CIN, const float* U, const float* shift,
                       const float* R, float* Y, int COUT, int relu, int y_first, hipStream_t st) {
  return R != nullptr ? launch_wino_r<TR, TCG, true>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, y_first, st)
                      : launch_wino_r<TR, TCG, false>(X, B, H, W, CIN, U, shift, R, Y, COUT, relu, y_first, st);
}

}  // namespace pa

extern "C" {

#if PA_WINO_STAMP
int pa_wino_read_stamps(unsigned long long* host, int zero) {
  const size_t n = sizeof(unsigned long long) * pa::STAMP_WG * 4 * pa::STAMP_IT * pa::STAMP_PH;
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pa::g_wino_stamps), n) != hipSuccess) return 1;
  if (zero) {
    void* d = nullptr;
    (void)hipGetSymbolAddress(&d, HIP_SYMBOL(pa::g_wino_stamps));
    (void)hipMemset(d, 0, n);
  }
  return 0;
}
#endif

// conv3x3, stride 1, pad 1, via Winograd F(2x2,3x3): Y = [relu](conv(X) + shift [+ R]).
// U: G g G^T (BatchNorm scale folded) in the slab layout of weights.winograd_pack:
// [cout/32][cin/16][row = 32 xi + n][slot][4], xi = 4a + b.
int pa_conv3x3_wino_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                         const float* R, float* Y, int cout, int relu, int y_first, void* stream);

int pa_conv3x3_wino(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                    const float* R, float* Y, int cout, int relu, void* stream) {
  return pa_conv3x3_wino_rows(X, B, H, W, cin, U, shift, R, Y, cout, relu, 0, stream);
}

// the same convolution for the output rows y_first .. H - 1 only (y_first even; the rows above are somebody else's:
// pa_emb_forward gives the last two rows of a map whose height is 2 (mod 4) to this kernel and the rest to F(4x4))
int pa_conv3x3_wino_rows(const float* X, int B, int H, int W, int cin, const float* U, const float* shift,
                         const float* R, float* Y, int cout, int relu, int y_first, void* stream) {
  if (B <= 0 || y_first >= H) return 0;
  PA_REQUIRE(y_first >= 0 && y_first % 2 == 0, "pa_conv3x3_wino_rows: y_first must be even and >= 0 (got %d)", y_first);
  const int Hr = H - y_first;      // rows this launch covers
  PA_REQUIRE(cin % pa::WCB == 0 && cout % pa::W_BN == 0, "pa_conv3x3_wino: cin %% 16 and cout %% 32 required");
  PA_REQUIRE((long)H * W * (cin > cout ? cin : cout) * 4 < (1L << 28),
             "pa_conv3x3_wino: one image must be smaller than 256 MB");
  // `flops` = the direct convolution's (the reference's operation); the kernel executes 16/36 of them
  pa::ProfScope prof("k_conv3x3_wino", stream, 2.0 * 9 * cin * cout * (double)B * Hr * W,
                     4.0 * ((double)B * Hr * W * cin + (double)B * Hr * W * cout * (R ? 2 : 1) + 9.0 * cin * cout));
  hipStream_t st = (hipStream_t)stream;
  // workgroup tile = 2*TR x 32*TCG output pixels: pick the shape that pads the map the least (10 s chunks:
  // 80/40 rows -> 8 x 32, 20 rows -> 4 x 64, 10 rows x 125 columns -> 2 x 128; the 38-column maps of 3 s
  // segments would waste 70 % of a 128-wide tile and take 4 x 64 instead); ties go to the taller tile,
  // whose halo is relatively smaller
  auto padded = [&](int tr, int tcg) {
    return (long)pa::cdiv(Hr, 2 * tr) * 2 * tr * (long)pa::cdiv(W, 32 * tcg) * 32 * tcg;
  };
  const long a41 = padded(4, 1), a22 = padded(2, 2), a14 = padded(1, 4);
  static const bool wino32 = getenv("PA_WINO32") == nullptr || atoi(getenv("PA_WINO32")) != 0;   // A/B aid
  if (wino32 && cin == 32 && cout == 32 && y_first == 0 && a41 <= a22 && a41 <= a14) {
    const int rc = R != nullptr ? pa::launch_wino32<true>(X, B, H, W, U, shift, R, Y, relu, st)
                                : pa::launch_wino32<false>(X, B, H, W, U, shift, R, Y, relu, st);
    if (rc != 0) return rc;
  } else if (a41 <= a22 && a41 <= a14) pa::launch_wino<4, 1>(X, B, H, W, cin, U, shift, R, Y, cout, relu, y_first, st);
  else if (a22 <= a14) pa::launch_wino<2, 2>(X, B, H, W, cin, U, shift, R, Y, cout, relu, y_first, st);
  else pa::launch_wino<1, 4>(X, B, H, W, cin, U, shift, R, Y, cout, relu, y_first, st);
  PA_CHECK_LAUNCH("pa_conv3x3_wino");
  return 0;
}

}  // extern "C"
