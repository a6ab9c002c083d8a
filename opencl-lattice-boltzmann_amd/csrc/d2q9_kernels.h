// Device code of liblbm_hip.so: the fused D2Q9-BGK timestep for gfx950 (MI355X, CDNA4).
//
// What one launch of d2q9_step computes is the reference's accelerate_flow + timestep kernels
// (kernels.cl:9-53, 56-231) for a range of rows:
//   pull-stream gather with periodic wrap   kernels.cl:91-114
//   density / momenta / equilibria / BGK    kernels.cl:119-185
//   bounce-back on obstacle cells           kernels.cl:69,187-197
//   per-cell |j|/rho summed per workgroup   kernels.cl:198-229
// plus, on the row ny-2, the NEXT step's accelerate_flow applied while the row is still in
// registers (exact: accelerate_flow is local to a cell and runs directly before the next gather).
//
// Data layout in HBM (one grid): ROW-INTERLEAVED SoA.  Row y of the grid is one contiguous block
// of 9 plane-rows, f_k(x, y) at  y*row_stride + k*plane_stride + x  with plane_stride = the padded
// row length and row_stride = 9*plane_stride.  Each speed is still a unit-stride fp32 stream along
// x (coalesced float4 per lane), but the 9 streams a row update touches form 3 contiguous
// segments (rows y-1, y, y+1) instead of 9 far-apart planes — measured +12 % on 8192x8192 over
// the reference's plane-major SoA (d2q9-bgk.c:73), whose nine 256-MiB-apart streams collide in
// the HBM channel interleave.  The C ABI still speaks the reference's float[9][ny][nx]; upload
// converts with 2-D copies, download repacks on the device (pack_planes).  The obstacle mask is one byte per cell.  Every thread
// owns VEC=4 consecutive cells of one row: all 9 loads and all 9 stores of a wave are whole
// 1-KiB contiguous segments (64 lanes x 16 B).
//
// Kernels, by how many timesteps one launch advances (all inline the same collide_cell: bit-identical results):
//   d2q9_step   1   thread = 4 cells, workgroup = one 1024-cell tile; the bandwidth-bound baseline
//   d2q9_step2  2   wave = strip of 60 float4 lanes sweeping a chunk of rows; intermediate row window in registers
//   d2q9_step3  3   the same with two windows, both in LDS (two waves per SIMD)
//   d2q9_step4  4   a third window in registers and LDS leftovers; the default from 2M cells
//   d2q9_multi  <=8 1024-thread workgroup on an LDS-resident tile with redundant halo; launch-bound small grids
//
// Not a translation of kernels.cl: different work decomposition (float4 rows, grid-stride,
// wave64 shuffles), different arithmetic grouping (pairwise momentum differences, shared
// equilibrium terms), fused accelerate, byte mask, deterministic two-stage reduction.
// (The library is two translation units — lbm_hip.cpp and lbm_deep.cpp, see deep_instances.h —, so the kernels that are not
// templates are `static`: only the unit that launches them emits them.)
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lbm {

constexpr int kBlock = 256;  // 4 wave64 per workgroup

struct StepArgs {
  const float *src;           // plane 0 of the source grid
  float *dst;                 // plane 0 of the destination grid
  const uint8_t *mask;        // [rows][nx], non-zero = obstacle
  const float *south_src[3];  // row "y-1" of planes 2,5,6 for y == 0: the periodic wrap row (kernels.cl:91-93)
  const float *north_src[3];  // row "y+1" of planes 4,7,8 for y == rows-1
  float *partials;            // [gridDim.x] per-workgroup sums of |j|/rho
  unsigned long long plane_stride;  // floats between the 9 plane-rows of one grid row (>= nx)
  unsigned long long row_stride;    // floats between consecutive grid rows (>= 9*plane_stride)
  int nx, rows;               // row length, rows stored in this grid (a slab's halo rows included)
  int y_begin, y_count;       // rows processed: y_begin + r for r < y_split, y_begin2 + (r - y_split) above
  int y_split, y_begin2;      //   (two row ranges in one launch: the bottom and the top edge rows of a slab)
  int accel_row;              // local row that gets the next step's accelerate_flow, or -1
  float omega, aw1, aw2;      // relaxation; density*accel/9, density*accel/36 (kernels.cl:14-15)
};

// ---- small helpers ----------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
  // wave64 butterfly; every lane ends with the total
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void store4(float *p, float a, float b, float c, float d) {
  v4f v = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
  else *reinterpret_cast<v4f *>(p) = v;
}

// 16-byte load, non-temporal when the data is read exactly once (grids beyond the Infinity Cache)
template <bool NT>
__device__ __forceinline__ float4 ld4(const float *p) {
  if (NT) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4 *>(p);
}

// BGK collision of one cell on the gathered distributions g[0..8]; writes the new cell to out[].
// Returns |j|/rho for a fluid cell, 0 for an obstacle.  Arithmetic of kernels.cl:119-198 with
//  - momenta from pairwise differences (a cell at rest has exactly zero momentum in fp32),
//  - eq_k = w_k*(rho + 3 j_k + (1.5/rho)(3 j_k^2 - j^2)) regrouped around the shared term
//    c = rho - (1.5/rho) j^2.
template <bool MAY_BE_OBSTACLE = true>
__device__ __forceinline__ float collide_cell(const float (&g)[9], bool obstacle_in, float omega, float (&out)[9]) {
  // MAY_BE_OBSTACLE = false: the caller knows no cell of the wave is blocked; the bounce-back selects vanish
  // Every multiply-add below is spelled out (explicit fmaf, contraction off) so that all kernels that
  // inline this function — one step per launch, two steps per launch, any load mode — round identically:
  // their results are bit-for-bit equal whatever the surrounding code lets the compiler fuse.
#pragma clang fp contract(off)
  const bool obstacle = MAY_BE_OBSTACLE && obstacle_in;
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;
  float dens = g[0] + g[1];
  dens += g[2]; dens += g[3]; dens += g[4]; dens += g[5]; dens += g[6]; dens += g[7]; dens += g[8];
  const float densinv = __builtin_amdgcn_rcpf(dens);
  const float da = g[5] - g[7], db = g[8] - g[6];
  const float jx = (g[1] - g[3]) + (da + db);
  const float jy = (g[2] - g[4]) + (da - db);
  const float usq = __builtin_fmaf(jx, jx, jy * jy);
  const float h = 1.5f * densinv;                     // 0.5 * densinv * ic_sq
  const float c = __builtin_fmaf(-h, usq, dens);      // rho - (1.5/rho) j^2, shared by all nine equilibria
  const float h3 = 3.0f * h;
  const float jp = jx + jy, jm = jx - jy;
  const float ax = __builtin_fmaf(h3 * jx, jx, c), ay = __builtin_fmaf(h3 * jy, jy, c);
  const float ap = __builtin_fmaf(h3 * jp, jp, c), am = __builtin_fmaf(h3 * jm, jm, c);
  float eq[9];
  eq[0] = w0 * c;
  eq[1] = w1 * __builtin_fmaf(3.0f, jx, ax);
  eq[3] = w1 * __builtin_fmaf(-3.0f, jx, ax);
  eq[2] = w1 * __builtin_fmaf(3.0f, jy, ay);
  eq[4] = w1 * __builtin_fmaf(-3.0f, jy, ay);
  eq[5] = w2 * __builtin_fmaf(3.0f, jp, ap);
  eq[7] = w2 * __builtin_fmaf(-3.0f, jp, ap);
  eq[8] = w2 * __builtin_fmaf(3.0f, jm, am);
  eq[6] = w2 * __builtin_fmaf(-3.0f, jm, am);
  // fluid: relax towards equilibrium, f + omega*(eq - f); obstacle: the un-relaxed value leaves through
  // the opposite speed (kernels.cl:69 lookup table: 0<->0, 1<->3, 2<->4, 5<->7, 6<->8)
  out[0] = obstacle ? g[0] : __builtin_fmaf(omega, eq[0] - g[0], g[0]);
  out[1] = obstacle ? g[3] : __builtin_fmaf(omega, eq[1] - g[1], g[1]);
  out[2] = obstacle ? g[4] : __builtin_fmaf(omega, eq[2] - g[2], g[2]);
  out[3] = obstacle ? g[1] : __builtin_fmaf(omega, eq[3] - g[3], g[3]);
  out[4] = obstacle ? g[2] : __builtin_fmaf(omega, eq[4] - g[4], g[4]);
  out[5] = obstacle ? g[7] : __builtin_fmaf(omega, eq[5] - g[5], g[5]);
  out[6] = obstacle ? g[8] : __builtin_fmaf(omega, eq[6] - g[6], g[6]);
  out[7] = obstacle ? g[5] : __builtin_fmaf(omega, eq[7] - g[7], g[7]);
  out[8] = obstacle ? g[6] : __builtin_fmaf(omega, eq[8] - g[8], g[8]);
  return obstacle ? 0.0f : __builtin_amdgcn_sqrtf(usq) * densinv;
}

// accelerate_flow on one cell held in registers (kernels.cl:24-42)
__device__ __forceinline__ void accelerate_cell(float (&f)[9], bool obstacle, float aw1, float aw2) {
#pragma clang fp contract(off)
  if (!obstacle && (f[3] - aw1) > 0.0f && (f[6] - aw2) > 0.0f && (f[7] - aw2) > 0.0f) {
    f[1] += aw1; f[5] += aw2; f[8] += aw2;
    f[3] -= aw1; f[6] -= aw2; f[7] -= aw2;
  }
}

// workgroup sum -> partials[blockIdx.x]; fixed order, no atomics (deterministic)
__device__ __forceinline__ void block_store_partial(float v, float *partials) {
  __shared__ float wsum[kBlock / 64];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = wsum[0];
    for (int i = 1; i < kBlock / 64; i++) t += wsum[i];
    partials[blockIdx.x] = t;
  }
}

// gathered distributions g[k][v] -> collided cell values o[k][v]; returns the sum of |j|/rho of the 4 cells
// (tried and rejected, tools/ab.py: bounce-back as a wave-uniform fix-up branch after an obstacle-free
// collision — 14 % slower in d2q9_step3, 4-40 % slower in d2q9_step2: both g and o stay live across the branch)
__device__ __forceinline__ float collide4(const float (&g)[9][4], uint32_t m, float omega, bool accel, float aw1, float aw2,
                                          float (&o)[9][4]) {
  float tot = 0.f;
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const bool obst = ((m >> (8 * v)) & 0xffu) != 0;
    float gc[9], oc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) gc[k] = g[k][v];
    tot += collide_cell<true>(gc, obst, omega, oc);
#pragma unroll
    for (int k = 0; k < 9; k++) o[k][v] = oc[k];
  }
  // accelerate_flow touches one row of the grid: keep it out of the instruction stream of all other rows
  // (a real scalar branch on "any lane of the wave is on that row" instead of predicated code on every cell)
  if (__builtin_amdgcn_readfirstlane((int)(__ballot(accel) != 0ull))) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const bool obst = ((m >> (8 * v)) & 0xffu) != 0;
      float oc[9];
#pragma unroll
      for (int k = 0; k < 9; k++) oc[k] = o[k][v];
      if (accel) accelerate_cell(oc, obst, aw1, aw2);  // a wave may straddle two rows when nx/4 is not a multiple of 64
#pragma unroll
      for (int k = 0; k < 9; k++) o[k][v] = oc[k];
    }
  }
  return tot;
}

// ---- the step kernel ---------------------------------------------------------------------------
// How a thread obtains the x-1 / x+1 neighbours of its four cells (planes 1,5,8 stream from the
// west, 3,6,7 from the east).  HBM traffic is identical in all modes — 9 floats in + 9 floats out
// + 1 mask byte per cell; the modes differ in L1/LDS/VALU work only.
enum LoadMode {
  LM_SCALAR = 0,     // aligned float4 + one scalar load of the neighbour element (an L1 hit)
  LM_UNALIGNED = 1,  // one 4-byte-aligned 16-byte load at x-1 / x+1
  LM_DPP = 2,        // aligned float4 + wave64 DPP shift of the edge component; lanes 0/63 load the halo
  LM_LDS = 3,        // aligned float4 staged through a per-wave LDS row with a one-cell halo, read back shifted
};

struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };

__device__ __forceinline__ float dpp_from_lane_below(float v, float lane0_value) {
  // lane i receives lane i-1; lane 0 keeps lane0_value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_lane_above(float v, float lane63_value) {
  // lane i receives lane i+1; lane 63 keeps lane63_value
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane63_value), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

constexpr int kLdsRow = 256 + 8;  // per-wave staged row: [3 pad][W halo][256 cells][E halo][3 pad]

template <int VEC, bool NT, int LM>
__global__ __launch_bounds__(kBlock) void d2q9_step(const StepArgs a) {
  const unsigned tpr = (unsigned)a.nx / VEC;  // threads per row
  const unsigned total = tpr * (unsigned)a.y_count;
  const size_t ps = a.plane_stride;
  const size_t rs = a.row_stride;
  float tot_u = 0.0f;
  __shared__ float stage[(LM == LM_LDS) ? (kBlock / 64) * 6 * kLdsRow : 1];

  for (unsigned t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const unsigned r = t / tpr;
    const int x0 = (int)(t - r * tpr) * VEC;
    const int y = ((int)r < a.y_split) ? a.y_begin + (int)r : a.y_begin2 + ((int)r - a.y_split);
    const size_t row = (size_t)y * rs;
    // neighbour columns with periodic wrap (kernels.cl:99-102)
    const int xw = (x0 == 0) ? a.nx - 1 : x0 - 1;
    const int xe = (x0 + VEC >= a.nx) ? 0 : x0 + VEC;
    // source rows: same row for planes 0,1,3; south row for 2,5,6; north row for 4,7,8
    const float *rc = a.src + row;
    const bool at_south = (y == 0), at_north = (y == a.rows - 1);
    const float *r2 = at_south ? a.south_src[0] : a.src + 2 * ps + row - rs;
    const float *r5 = at_south ? a.south_src[1] : a.src + 5 * ps + row - rs;
    const float *r6 = at_south ? a.south_src[2] : a.src + 6 * ps + row - rs;
    const float *r4 = at_north ? a.north_src[0] : a.src + 4 * ps + row + rs;
    const float *r7 = at_north ? a.north_src[1] : a.src + 7 * ps + row + rs;
    const float *r8 = at_north ? a.north_src[2] : a.src + 8 * ps + row + rs;
    const float *r1 = rc + 1 * ps, *r3 = rc + 3 * ps;

    float g[9][VEC];
    bool obst[VEC];
    uint32_t mask_word = 0;
    if constexpr (VEC == 4) {
      // all loads are issued before any use
      const float4 c0 = ld4<NT && LM != LM_SCALAR>(rc + x0);
      const float4 c2 = ld4<NT && LM != LM_SCALAR>(r2 + x0);
      const float4 c4 = ld4<NT && LM != LM_SCALAR>(r4 + x0);
      const uint32_t m = *reinterpret_cast<const uint32_t *>(a.mask + (size_t)y * a.nx + x0);
      g[0][0] = c0.x; g[0][1] = c0.y; g[0][2] = c0.z; g[0][3] = c0.w;
      g[2][0] = c2.x; g[2][1] = c2.y; g[2][2] = c2.z; g[2][3] = c2.w;
      g[4][0] = c4.x; g[4][1] = c4.y; g[4][2] = c4.z; g[4][3] = c4.w;
      mask_word = m;
      obst[0] = (m & 0xffu) != 0; obst[1] = (m & 0xff00u) != 0;
      obst[2] = (m & 0xff0000u) != 0; obst[3] = (m & 0xff000000u) != 0;
      if constexpr (LM == LM_UNALIGNED) {
        // x0 == 0 / x0+4 == nx wrap around the row: those two threads per row take the scalar path
        const bool wrap_w = (x0 == 0), wrap_e = (x0 + 4 >= a.nx);
        const int ow = wrap_w ? 0 : -1, oe = wrap_e ? 0 : 1;
        f4u w1 = *reinterpret_cast<const f4u *>(r1 + x0 + ow), w5 = *reinterpret_cast<const f4u *>(r5 + x0 + ow),
            w8 = *reinterpret_cast<const f4u *>(r8 + x0 + ow);
        f4u e3 = *reinterpret_cast<const f4u *>(r3 + x0 + oe), e6 = *reinterpret_cast<const f4u *>(r6 + x0 + oe),
            e7 = *reinterpret_cast<const f4u *>(r7 + x0 + oe);
        if (wrap_w) {
          w1 = f4u{r1[xw], w1.x, w1.y, w1.z}; w5 = f4u{r5[xw], w5.x, w5.y, w5.z}; w8 = f4u{r8[xw], w8.x, w8.y, w8.z};
        }
        if (wrap_e) {
          e3 = f4u{e3.y, e3.z, e3.w, r3[xe]}; e6 = f4u{e6.y, e6.z, e6.w, r6[xe]}; e7 = f4u{e7.y, e7.z, e7.w, r7[xe]};
        }
        g[1][0] = w1.x; g[1][1] = w1.y; g[1][2] = w1.z; g[1][3] = w1.w;
        g[5][0] = w5.x; g[5][1] = w5.y; g[5][2] = w5.z; g[5][3] = w5.w;
        g[8][0] = w8.x; g[8][1] = w8.y; g[8][2] = w8.z; g[8][3] = w8.w;
        g[3][0] = e3.x; g[3][1] = e3.y; g[3][2] = e3.z; g[3][3] = e3.w;
        g[6][0] = e6.x; g[6][1] = e6.y; g[6][2] = e6.z; g[6][3] = e6.w;
        g[7][0] = e7.x; g[7][1] = e7.y; g[7][2] = e7.z; g[7][3] = e7.w;
      } else {
        const float4 c1 = ld4<NT && LM != LM_SCALAR>(r1 + x0);
        const float4 c3 = ld4<NT && LM != LM_SCALAR>(r3 + x0);
        const float4 c5 = ld4<NT && LM != LM_SCALAR>(r5 + x0);
        const float4 c6 = ld4<NT && LM != LM_SCALAR>(r6 + x0);
        const float4 c7 = ld4<NT && LM != LM_SCALAR>(r7 + x0);
        const float4 c8 = ld4<NT && LM != LM_SCALAR>(r8 + x0);
        float w1, w5, w8, e3, e6, e7;
        if constexpr (LM == LM_SCALAR) {
          w1 = r1[xw]; w5 = r5[xw]; w8 = r8[xw];
          e3 = r3[xe]; e6 = r6[xe]; e7 = r7[xe];
        } else if constexpr (LM == LM_DPP) {
          // requires whole waves inside one row (tpr % 64 == 0): lane i-1 / i+1 hold x0-4 / x0+4
          const int lane = threadIdx.x & 63;
          float h0 = 0.f, h1 = 0.f, h2 = 0.f;
          if (lane == 0 || lane == 63) {
            // one masked load per plane pair: lane 0 fetches the west halo, lane 63 the east halo
            const bool lo = (lane == 0);
            h0 = lo ? r1[xw] : r3[xe];
            h1 = lo ? r5[xw] : r6[xe];
            h2 = lo ? r8[xw] : r7[xe];
          }
          w1 = dpp_from_lane_below(c1.w, h0); w5 = dpp_from_lane_below(c5.w, h1); w8 = dpp_from_lane_below(c8.w, h2);
          e3 = dpp_from_lane_above(c3.x, h0); e6 = dpp_from_lane_above(c6.x, h1); e7 = dpp_from_lane_above(c7.x, h2);
        } else {  // LM_LDS
          const int lane = threadIdx.x & 63;
          float *st = stage + (threadIdx.x >> 6) * 6 * kLdsRow;
          float *s1 = st, *s5 = st + kLdsRow, *s8 = st + 2 * kLdsRow, *s3 = st + 3 * kLdsRow, *s6 = st + 4 * kLdsRow,
                *s7 = st + 5 * kLdsRow;
          const int o = 4 + 4 * lane;
          *reinterpret_cast<float4 *>(s1 + o) = c1; *reinterpret_cast<float4 *>(s5 + o) = c5;
          *reinterpret_cast<float4 *>(s8 + o) = c8; *reinterpret_cast<float4 *>(s3 + o) = c3;
          *reinterpret_cast<float4 *>(s6 + o) = c6; *reinterpret_cast<float4 *>(s7 + o) = c7;
          if (lane == 0) { s1[3] = r1[xw]; s5[3] = r5[xw]; s8[3] = r8[xw]; }
          if (lane == 63) { s3[4 + 256] = r3[xe]; s6[4 + 256] = r6[xe]; s7[4 + 256] = r7[xe]; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          w1 = s1[o - 1]; w5 = s5[o - 1]; w8 = s8[o - 1];
          e3 = s3[o + 4]; e6 = s6[o + 4]; e7 = s7[o + 4];
          __builtin_amdgcn_wave_barrier();
        }
        g[1][0] = w1;   g[1][1] = c1.x; g[1][2] = c1.y; g[1][3] = c1.z;   // from x-1
        g[5][0] = w5;   g[5][1] = c5.x; g[5][2] = c5.y; g[5][3] = c5.z;
        g[8][0] = w8;   g[8][1] = c8.x; g[8][2] = c8.y; g[8][3] = c8.z;
        g[3][0] = c3.y; g[3][1] = c3.z; g[3][2] = c3.w; g[3][3] = e3;     // from x+1
        g[6][0] = c6.y; g[6][1] = c6.z; g[6][2] = c6.w; g[6][3] = e6;
        g[7][0] = c7.y; g[7][1] = c7.z; g[7][2] = c7.w; g[7][3] = e7;
      }
    } else {
      g[0][0] = rc[x0]; g[1][0] = r1[xw]; g[2][0] = r2[x0]; g[3][0] = r3[xe]; g[4][0] = r4[x0];
      g[5][0] = r5[xw]; g[6][0] = r6[xe]; g[7][0] = r7[xe]; g[8][0] = r8[xw];
      obst[0] = a.mask[(size_t)y * a.nx + x0] != 0;
    }

    float o[9][VEC];
    const bool accel_here = (y == a.accel_row);
    if constexpr (VEC == 4) {
      tot_u += collide4(g, mask_word, a.omega, accel_here, a.aw1, a.aw2, o);
    } else {
      float gc[9], oc[9];
#pragma unroll
      for (int k = 0; k < 9; k++) gc[k] = g[k][0];
      tot_u += collide_cell(gc, obst[0], a.omega, oc);
      if (accel_here) accelerate_cell(oc, obst[0], a.aw1, a.aw2);
#pragma unroll
      for (int k = 0; k < 9; k++) o[k][0] = oc[k];
    }

    float *d = a.dst + row + x0;
    if constexpr (VEC == 4) {
#pragma unroll
      for (int k = 0; k < 9; k++) store4<NT>(d + k * ps, o[k][0], o[k][1], o[k][2], o[k][3]);
    } else {
#pragma unroll
      for (int k = 0; k < 9; k++) d[k * ps] = o[k][0];
    }
  }
  block_store_partial(tot_u, a.partials);
}

// ---- peer-halo transport: push kernel + flag words ------------------------------------------------
// A slab's edge rows go straight into the ring neighbours' halo rows (peer-mapped memory: the same process, another
// device with peer access, or another process through HIP IPC), followed by a sequence number in the neighbour's
// flag word.  Protocol (csrc/lbm_hip.cpp, exchange_halos): the consumer waits for flag >= seq before the launch that
// reads the halo rows; that the producer may overwrite them again follows from the data dependencies of the ring
// (its next push comes after its next edge launch, which waited for this consumer's previous push, which came
// after the consumer's last reader of those rows).
struct PushArgs {
  const float *src_lo, *src_hi;   // this slab's bottom / top edge rows (halo_depth rows each, contiguous)
  float *dst_lo, *dst_hi;         // the south neighbour's top halo rows / the north neighbour's bottom halo rows
  unsigned long long n4;          // float4 per block
  uint32_t *flag_lo, *flag_hi;    // the neighbours' flag words for pushes arriving from this side
  uint32_t seq;                   // sequence number of this exchange
  unsigned *ticket;               // workgroups done (reset by the last one)
  int release;                    // 1: release fences around the ticket (release_pushed), 0: drained write-through stores
};

__device__ __forceinline__ void flag_store_system(uint32_t *flag, uint32_t v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t flag_load_system(const uint32_t *flag) {
  return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// WRITE-THROUGH stores (sc0 sc1: system scope, nothing stays dirty in this XCD's L2).  The pushed halo rows are made
// visible by these stores themselves plus each storing wave's s_waitcnt vmcnt(0) — not by release fences: a fence writes
// back (and, as __threadfence_system, invalidates) the whole L2 of the XCD, which the interior launch running beside
// the push is busy filling (first version, one fence per thread: 8192x1024 slab 44.1 instead of 38.1 us/step).
// Written as two 8-byte system-scope relaxed atomic stores, which the compiler turns into global_store_dwordx2 sc0 sc1,
// NOT as inline assembly: the hazard recognizer does not look inside an asm block, and a hand-written
// global_store_dwordx4 there went wrong twice — issued right behind the v_readfirstlane that produced its SGPR base (a
// vector-memory instruction may read a VALU-written SGPR only five wait states later: stale base, GPU memory fault) and
// right in front of a VALU instruction that re-used its data registers (a store of more than 8 bytes reads its data
// late: seven of eight pushed rows arrived corrupted).
typedef __attribute__((address_space(1))) unsigned long long global_u64;  // a pointer known to be global memory: global_store, not flat_store
__device__ __forceinline__ void store4_through(float *p, v4f v) {
  global_u64 *q = (global_u64 *)(unsigned long long)p;
  const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// a value every lane holds alike, moved to scalar registers (what the compiler cannot prove for values loaded through
// the late argument pointer)
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
// the same from (wave-uniform base pointer) + (the lane's 32-bit byte offset): the SGPR-base addressing form
__device__ __forceinline__ void store4_through_sbase(float *uniform_base, unsigned byte_off, v4f v) {
  store4_through(reinterpret_cast<float *>(reinterpret_cast<char *>(uniform_base) + byte_off), v);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
// What a pushing wave does between its last write-through store and its arrival at the ticket.  Two forms, chosen per slab
// by the host (HaloPeer::release, option "push_release"):
//   formal (1)  a RELEASE FENCE at system scope (buffer_wbl2 sc0 sc1 + s_waitcnt): the pushed rows happen-before the
//               ticket increment in the sense of the HSA memory model, whatever the cache policy of the mapping they went
//               through.  The default whenever a ring neighbour lives on another device (xGMI), where nothing but the
//               model's guarantees has ever been exercised.
//   drain (0)   s_waitcnt vmcnt(0) only: enough for write-through (sc0 sc1) stores, whose acknowledgement means the data has
//               left this XCD's L2 — measured and soak-tested between slabs, processes (HIP IPC) on ONE device; it spares the
//               L2 write-back that the fence also performs for the interior launch's dirty lines.
__device__ __forceinline__ void release_pushed(int formal) {
  if (formal) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  else drain_stores();
}

// The last workgroup of `expected` to arrive publishes the sequence number.  Precondition: every wave of the calling
// workgroup has drained its write-through stores (drain_stores) before the barrier in here.
// `formal`: the publisher joins the arrivals' release fences to the flag stores with an acquire-release fence of its own
// (ticket read -> acquire, flag store <- release), so that pushed rows -> ticket -> flag is one happens-before chain that
// ends in the consumer's acquire (halo_wait's fence + the kernel-start acquire of the launch that reads the halo rows).
__device__ __forceinline__ void publish_when_last(unsigned *ticket, unsigned expected, uint32_t *flag_a, uint32_t *flag_b, uint32_t seq,
                                                  int formal = 0) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == expected - 1) {
      if (formal) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
        drain_stores();  // (the waitcnt pass drops the fence's own s_waitcnt behind buffer_wbl2 when it sees no store pending)
      }
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next user starts after this kernel has ended
      if (flag_a) flag_store_system(flag_a, seq);
      if (flag_b) flag_store_system(flag_b, seq);
    }
  }
}

static __global__ __launch_bounds__(kBlock) void halo_push(const PushArgs a) {
  const unsigned half = gridDim.x >> 1;  // first half of the workgroups: bottom rows, second half: top rows
  const bool hi = blockIdx.x >= half;
  const v4f *src = reinterpret_cast<const v4f *>(hi ? a.src_hi : a.src_lo);
  float *dst = hi ? a.dst_hi : a.dst_lo;
  const unsigned b = hi ? blockIdx.x - half : blockIdx.x;
  for (size_t i = (size_t)b * kBlock + threadIdx.x; i < a.n4; i += (size_t)half * kBlock) store4_through(dst + 4 * i, src[i]);
  release_pushed(a.release);
  publish_when_last(a.ticket, gridDim.x, a.flag_lo, a.flag_hi, a.seq, a.release);
}

// Consumer side: one wave, lanes 0 and 1 poll the two flag words until both have reached `seq`.  The spin is
// BOUNDED (timeout in 100-MHz ticks of s_memrealtime): a neighbour that never arrives sets the error word, which
// lbm_sync reports, instead of hanging the GPU.  The launch that follows starts with the usual kernel-start acquire.

// What a slab's kernels need to know about its ring neighbours, device-resident (filled once the ring is connected):
// as kernel arguments these pointers stayed live in SGPRs to the end of the kernel (d2q9_multi: 92 instead of 76, which
// costs the 16x16 / 16x8 tiles their second workgroup per CU — more than 80 SGPRs admit 28 waves per CU, not 32).
struct HaloPeer {
  float *push[2][2];               // [side][grid]: side 0 = the south neighbour's top halo rows (this slab's bottom edge rows go
                                   // there), side 1 = the north neighbour's bottom halo rows (this slab's top edge rows)
  uint32_t *flag_lo, *flag_hi;     // the neighbours' flag words for pushes arriving from this slab
  unsigned *ticket;
  const uint32_t *wait_flags;      // this slab's own flag words
  uint32_t *wait_err;
  unsigned long long wait_ticks;
  int push_rows;                   // halo depth H
  int row_lo0, row_hi0;            // stored rows of the first bottom edge row / the first top edge row (the rows that are pushed)
  int release;                     // how a pushing wave orders its rows before the ticket: see release_pushed
};

// The kernel's own argument block, re-read where it is used: fields needed only at the very start or the very end of
// a long kernel otherwise sit in SGPRs all the way through (the asm hides that this is the preloaded argument copy).
template <class Args>
__device__ __forceinline__ const Args *late_args() {
  unsigned long long p = (unsigned long long)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return reinterpret_cast<const Args *>(p);
}

// bounded spin on one flag word: gives up after `timeout` ticks of s_memrealtime (100 MHz) and raises the error word.
// ONE timeout per run, not one per launch set: a wait that finds the error word already raised — by an earlier wait of
// this slab that timed out, in this launch or in any launch before it — falls through at once, so the launch sets still
// queued behind a neighbour that died drain at kernel speed instead of spinning (sets) x (timeout); lbm_sync reports
// the failure and the context refuses further work.
__device__ __forceinline__ void spin_on_flag(const uint32_t *f, uint32_t seq, uint32_t *err, unsigned long long timeout) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((int32_t)(flag_load_system(f) - seq) < 0) {
    __builtin_amdgcn_s_sleep(4);
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout) {
      atomicOr(err, 1u);
      break;
    }
  }
}
// lanes 0 and 1 of a workgroup on the slab's two flag words (see halo_wait)
__device__ __forceinline__ void spin_on_flags(const uint32_t *flags, uint32_t seq, uint32_t *err, unsigned long long timeout) {
  if (threadIdx.x < 2) spin_on_flag(flags + threadIdx.x, seq, err, timeout);
}

static __global__ void halo_wait(const uint32_t *flags, uint32_t seq, uint32_t *err, unsigned long long timeout) {
  spin_on_flags(flags, seq, err, timeout);
  // The acquire that pairs with the publisher's release.  What actually makes the halo rows readable is the boundary
  // behind this kernel: the rows live in THIS device's memory and were written by another agent past this device's
  // caches, so a reader needs its L1 / L2 copies of those lines dropped — which the kernel-start acquire of the next
  // launch on the stream does on every XCD that runs it (a fence here reaches only the XCD this one wave runs on;
  // that is why polling inside the consuming kernel, halo_sync 2, is refused between devices).
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}


// the last of `expected` waves (each after draining its own write-through stores) publishes the sequence number
__device__ __forceinline__ void publish_wave_when_last(const HaloPeer *pp, unsigned expected, uint32_t seq) {
  if ((threadIdx.x & 63) == 0) {
    const unsigned t = __hip_atomic_fetch_add(pp->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == expected - 1) {
      if (pp->release) {  // see publish_when_last
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
        drain_stores();
      }
      __hip_atomic_store(pp->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag_store_system(pp->flag_lo, seq);
      flag_store_system(pp->flag_hi, seq);
    }
  }
}

// ---- two timesteps per launch (temporal blocking) ------------------------------------------------
// One wave64 = one work unit: a strip of 64 float4-lanes sweeping `chunk_rows` grid rows upward.  Per
// row it (1) gathers the source rows and collides once (the intermediate state I, step t+1) and keeps I
// in REGISTERS — every value of I has exactly one consumer, and that consumer is this lane or the lane
// next to it — then (2) gathers from I (own registers for the y-shifts, wave64 DPP for the x-shifts),
// collides again (step t+2) and stores.  The grid is read once and written once per TWO steps:
// ~38 B instead of 72 B of HBM traffic per lattice update.  Lane 0 and the lane after the strip's last
// output lane are halo lanes: they compute I for the neighbouring strips' edge cells (redundantly) and
// produce no output, so waves never communicate: no LDS, no barrier, no atomics.  Redundant work: 2 of
// <= 64 lanes per row, 2 of chunk_rows+2 rows per unit.
// Rows and columns wrap periodically (kernels.cl:91-102); used for a grid held by one slab.
struct Step2Args {
  const float *src;
  float *dst;
  const uint8_t *mask;
  float *partials1;  // [units] sum of |j|/rho of step t+1 over the unit's owned cells
  float *partials2;  // [units] same for step t+2
  unsigned long long plane_stride, row_stride;
  int nx, ny;
  int strips, lanes_out;       // strips per row; output lanes per strip (<= 62); lane l owns float4 column s*lanes_out + l-1
  const int *chunk_start;      // [nchunks+1] first row of every chunk (chunks get shorter towards the end of a band)
  int nchunks;                 // units = strips * nchunks
  int nbands, units_per_band;  // workgroup b works on unit (b % nbands)*units_per_band + b / nbands
  int skip_chunk;              // chunk index whose units do nothing (-1: none); slab mode's edge launch skips the interior
  int accel_row;               // stored row that holds global row ny-2 (-1: none)
  int accel_row_b;             // a second stored copy of it (a slab that is its own ring neighbour), else -1
  int accel_next;              // apply the following step's accelerate_flow to the output row ny-2
  float omega, aw1, aw2;
  // Compact launch sets (peer transport; PUSH instantiations of d2q9_step3/3p/4/4p): ONE launch per launch set.  Its
  // first edge_units workgroups work through the edge schedule (the halo_depth rows at either end of the slab — what
  // the ring neighbours need), the others through the interior schedule above; an edge unit stores its output rows a
  // second time, write-through, into the neighbour's halo rows, and the last edge wave to finish raises the neighbours'
  // flag words to seq (peer_mode bit 0; without it — the last launch set of a run — the same launch, no push).
  const int *edge_chunk_start;   // chunk table of the edge schedule {bottom edge, (interior: skipped), top edge, ...}
  int edge_nchunks, edge_units, edge_skip;
  int edge_partial_off;          // edge unit u keeps its velocity sums in slot edge_partial_off + u
  const struct HaloPeer *peer;
  uint32_t seq, wait_seq;        // peer_mode bit 1 (option halo_sync = 2): the edge waves — the only readers of halo rows —
                                 // poll the slab's own flag words for wait_seq before their first load (bounded spin)
  int peer_mode, peer_buf;       // peer_buf: which of the neighbours' two grids the pushed rows go to
  // d2q9_deep / d2q9_deep_twin at the depths that have a kernel of their own: bit (r & 63) of word clean_bits[strip*clean_words
  // + (r >> 6)] is set when stored row r holds a blocked cell inside that strip's 64 lanes (strip_row_bits); a wave whose rows
  // are all clear runs the sweep without obstacle handling (deep_sweep<..., FREE>).  nullptr: no map, every wave looks.
  const unsigned long long *clean_bits;
  int clean_words;
  // d2q9_deep / d2q9_deep_twin, one-round schedules with a few strips that are slow because they hold blocked cells in most
  // rows (a cavity's wall strips): such a strip appears TWICE among the `strips` (virtual) strips of the interior decode, each
  // copy working on one half of every chunk (pair) of the table — half the rows per wave, so that the launch does not end
  // with the wall strips' waves.  vmap[2v] = the real strip of virtual strip v, vmap[2v+1] = its table t; vtab[2(t*nchunks +
  // chunk)], [.. + 1] = first row and end of that chunk in table t (table 0: the plain schedule).  nullptr: strips are strips.
  const int *vmap;
  const int *vtab;
  int strips_edge;             // real strips per row: what the EDGE units of a compact launch set are numbered by
};

// Which work unit a workgroup has (compact launch sets: edge schedule first).  All wave-uniform.
struct UnitSel {
  const int *chunk_start;
  int bid, nbands, units_per_band, skip, partial_off;
  bool edge;
};
template <bool PUSH>
__device__ __forceinline__ UnitSel select_unit(const Step2Args &a, int edge_wgs) {
  UnitSel u{a.chunk_start, (int)blockIdx.x, a.nbands, a.units_per_band, a.skip_chunk, 0, false};
  if constexpr (PUSH) {
    if (u.bid < edge_wgs) {
      u.edge = true;
      u.chunk_start = a.edge_chunk_start;
      u.nbands = 1;
      u.units_per_band = edge_wgs;
      u.skip = a.edge_skip;
      u.partial_off = a.edge_partial_off;
    } else {
      u.bid -= edge_wgs;
    }
  }
  return u;
}

// An edge unit's output rows a second time: into the ring neighbour's halo rows, write-through.  Done after the sweep,
// from the rows the wave has just stored (every lane re-reads exactly what it wrote itself: program order makes its own
// stores visible to it), so that the row loop of the PUSH instantiations is the row loop of the plain kernels — pushing
// from inside the loop cost the four-step kernel 47-66 spilled registers.  The rows of an edge chunk are, by
// construction, halo_depth rows at the bottom (stored rows row_lo0 ..) or at the top (row_hi0 ..) of the slab; a chunk
// that is neither raises the slab's error word instead of storing anywhere (lbm_sync reports it).
__device__ __forceinline__ void push_chunk(const Step2Args *la, int ys, int ye, int xcol, bool owner) {
  drain_stores();
  const HaloPeer *pp = uniform_ptr(la->peer);
  const int hi0 = __builtin_amdgcn_readfirstlane(pp->row_hi0);
  const int lo0 = __builtin_amdgcn_readfirstlane(pp->row_lo0);
  const int rows = __builtin_amdgcn_readfirstlane(pp->push_rows);
  const int buf = __builtin_amdgcn_readfirstlane(la->peer_buf) & 1;
  const int side = ys >= hi0 ? 1 : 0;
  const int base = side ? hi0 : lo0;
  if (ys < base || ye > base + rows) {
    if ((threadIdx.x & 63) == 0) atomicOr(pp->wait_err, 2u);
    return;
  }
  const size_t rs = la->row_stride, ps = la->plane_stride;
  const float *own = la->dst + xcol;
  float *peer = pp->push[side][buf] + xcol;
  if (owner) {
    for (int y = ys; y < ye; y++) {
      const float *src = own + (size_t)y * rs;
      float *dst = peer + (size_t)(y - base) * rs;
      v4f v[9];
#pragma unroll
      for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const v4f *>(src + k * ps);
#pragma unroll
      for (int k = 0; k < 9; k++) store4_through(dst + k * ps, v[k]);
    }
  }
  release_pushed(__builtin_amdgcn_readfirstlane(pp->release));
}

struct RowLoads {
  float4 c[9];
  float h0, h1, h2;  // halo elements (lane 0: west of planes 1,5,8; last lane: east of planes 3,6,7)
  uint32_t m;
};

template <bool NTL>
__device__ __forceinline__ float4 load4(const float *p) {
  if (NTL) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const float4 *>(p);
}

template <bool NTL>
__device__ __forceinline__ void issue_row_loads(const Step2Args &a, int r, int xcol, int xhalo_w, int xhalo_e, int lane,
                                                RowLoads &in) {
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int r_s = (r == 0) ? a.ny - 1 : r - 1;
  const int r_n = (r == a.ny - 1) ? 0 : r + 1;
  const float *Rc = a.src + (size_t)r * rs, *Rs = a.src + (size_t)r_s * rs, *Rn = a.src + (size_t)r_n * rs;
  in.c[0] = load4<NTL>(Rc + xcol);
  in.c[1] = load4<NTL>(Rc + 1 * ps + xcol);
  in.c[3] = load4<NTL>(Rc + 3 * ps + xcol);
  in.c[2] = load4<NTL>(Rs + 2 * ps + xcol);
  in.c[5] = load4<NTL>(Rs + 5 * ps + xcol);
  in.c[6] = load4<NTL>(Rs + 6 * ps + xcol);
  in.c[4] = load4<NTL>(Rn + 4 * ps + xcol);
  in.c[7] = load4<NTL>(Rn + 7 * ps + xcol);
  in.c[8] = load4<NTL>(Rn + 8 * ps + xcol);
  in.m = *reinterpret_cast<const uint32_t *>(a.mask + (size_t)r * a.nx + xcol);
  in.h0 = in.h1 = in.h2 = 0.f;
  if (lane == 0 || lane == 63) {
    const bool lo = (lane == 0);
    in.h0 = lo ? Rc[1 * ps + xhalo_w] : Rc[3 * ps + xhalo_e];
    in.h1 = lo ? Rs[5 * ps + xhalo_w] : Rs[6 * ps + xhalo_e];
    in.h2 = lo ? Rn[8 * ps + xhalo_w] : Rn[7 * ps + xhalo_e];
  }
}

// The same loads written as (wave-uniform row pointer) + (32-bit unsigned byte offset of the lane): the shape the
// SGPR-base addressing mode of global_load takes — no 64-bit vector address per access, six VGPRs fewer.  Worth
// nothing in d2q9_step3 (tools/ab.py), but d2q9_step4 sits at the 256-register limit, where every register
// that is not spilled counts.
__device__ __forceinline__ const float *at_byte(const float *uniform_base, unsigned byte_off) {
  return reinterpret_cast<const float *>(reinterpret_cast<const char *>(uniform_base) + byte_off);
}
template <bool NTL>
__device__ __forceinline__ void issue_row_loads_sbase(const Step2Args &a, int r, int xcol, int xhalo_w, int xhalo_e, int lane,
                                                      RowLoads &in) {
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int r_s = (r == 0) ? a.ny - 1 : r - 1;
  const int r_n = (r == a.ny - 1) ? 0 : r + 1;
  const float *Rc = a.src + (size_t)r * rs, *Rs = a.src + (size_t)r_s * rs, *Rn = a.src + (size_t)r_n * rs;
  unsigned xb = (unsigned)xcol * 4u;
  // keeps the 32->64-bit extension of the offset next to its uses: instruction selection works per basic block and
  // only then recognises base + zext(offset) as the SGPR-base addressing mode
  asm volatile("" : "+v"(xb));
  in.c[0] = load4<NTL>(at_byte(Rc, xb));
  in.c[1] = load4<NTL>(at_byte(Rc + 1 * ps, xb));
  in.c[3] = load4<NTL>(at_byte(Rc + 3 * ps, xb));
  in.c[2] = load4<NTL>(at_byte(Rs + 2 * ps, xb));
  in.c[5] = load4<NTL>(at_byte(Rs + 5 * ps, xb));
  in.c[6] = load4<NTL>(at_byte(Rs + 6 * ps, xb));
  in.c[4] = load4<NTL>(at_byte(Rn + 4 * ps, xb));
  in.c[7] = load4<NTL>(at_byte(Rn + 7 * ps, xb));
  in.c[8] = load4<NTL>(at_byte(Rn + 8 * ps, xb));
  in.m = *reinterpret_cast<const uint32_t *>(a.mask + (size_t)r * a.nx + xcol);
  in.h0 = in.h1 = in.h2 = 0.f;
  if (lane == 0 || lane == 63) {
    const bool lo = (lane == 0);
    in.h0 = lo ? Rc[1 * ps + xhalo_w] : Rc[3 * ps + xhalo_e];
    in.h1 = lo ? Rs[5 * ps + xhalo_w] : Rs[6 * ps + xhalo_e];
    in.h2 = lo ? Rn[8 * ps + xhalo_w] : Rn[7 * ps + xhalo_e];
  }
}

// shifts a float4-per-lane plane by one cell: result[v] = value at x-1 (west) or x+1 (east)
__device__ __forceinline__ void shift_from_west(const float (&p)[4], float halo, float (&out)[4]) {
  out[0] = dpp_from_lane_below(p[3], halo); out[1] = p[0]; out[2] = p[1]; out[3] = p[2];
}
__device__ __forceinline__ void shift_from_east(const float (&p)[4], float halo, float (&out)[4]) {
  out[0] = p[1]; out[1] = p[2]; out[2] = p[3]; out[3] = dpp_from_lane_above(p[0], halo);
}

__device__ __forceinline__ void unpack4(const float4 &v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }

// step t+1 of one row from the loaded source values
__device__ __forceinline__ float first_step_row(const Step2Args &a, const RowLoads &in, int r, float (&I)[9][4]) {
  float g[9][4], c[4];
  unpack4(in.c[0], g[0]); unpack4(in.c[2], g[2]); unpack4(in.c[4], g[4]);
  unpack4(in.c[1], c); shift_from_west(c, in.h0, g[1]);
  unpack4(in.c[5], c); shift_from_west(c, in.h1, g[5]);
  unpack4(in.c[8], c); shift_from_west(c, in.h2, g[8]);
  unpack4(in.c[3], c); shift_from_east(c, in.h0, g[3]);
  unpack4(in.c[6], c); shift_from_east(c, in.h1, g[6]);
  unpack4(in.c[7], c); shift_from_east(c, in.h2, g[7]);
  // the intermediate row ny-2 receives step t+2's accelerate_flow before it is streamed (kernels.cl:9-53)
  return collide4(g, in.m, a.omega, r == a.accel_row || r == a.accel_row_b, a.aw1, a.aw2, I);
}

// NTL: 0 = plain source loads, 1 = all non-temporal, 2 = non-temporal except for the two intermediate rows at
// either end of a chunk, whose source rows the neighbouring chunk reads as well (they should stay in L2)
template <bool NT, int NTL = 0>
__global__ __launch_bounds__(64) void d2q9_step2(const Step2Args a) {
  const int lane = threadIdx.x;
  // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one, MI355X_MICROARCH.md), so
  // band (b % nbands) of contiguous units stays on one XCD: neighbouring strips and chunks — which
  // re-read each other's edge lines and boundary rows — then share that XCD's L2.  Speed only.
  const int band = blockIdx.x % a.nbands, slot = blockIdx.x / a.nbands;
  if (slot >= a.units_per_band) return;
  const int unit = band * a.units_per_band + slot;
  const int chunk = unit / a.strips, strip = unit - chunk * a.strips;
  const int ys = a.chunk_start[chunk];
  const int ye = a.chunk_start[chunk + 1];
  if (ys >= ye || chunk == a.skip_chunk) {  // padding chunk of a short band / not this launch's business
    if (lane == 0) a.partials1[unit] = a.partials2[unit] = 0.f;
    return;
  }
  const int q4 = a.nx >> 2;                              // float4 columns per row
  const int qcol = strip * a.lanes_out + lane - 1;        // this lane's float4 column, unwrapped
  const bool owner = (lane >= 1) && (lane <= a.lanes_out) && (qcol < q4);
  int qw = qcol % q4;
  if (qw < 0) qw += q4;
  const int xcol = qw * 4;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 4 >= a.nx) ? 0 : xcol + 4;
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };

  // Neighbouring chunks sweep in OPPOSITE directions (even chunks upward, odd ones downward): two chunks
  // that share a boundary then reach it at the same time — both at their start or both at their end — so
  // the rows they both read (each chunk computes one intermediate row beyond either end) are fetched from
  // HBM once and found in the XCD's L2 the second time.  The direction only changes which three planes
  // travel with the older row of the register window (2,5,6 upward / 4,8,7 downward): wave-uniform selects.
  const bool up = __builtin_amdgcn_readfirstlane((int)((chunk & 1) == 0)) != 0;
  const int n = ye - ys;
  const int d = up ? 1 : -1;
  const int r0 = up ? ys - 1 : ye;  // k-th intermediate row computed: r0 + k*d, k = 0 .. n+1; rows k = 1..n are owned

  float sum1 = 0.f, sum2 = 0.f;
  float trail[3][4];  // of the oldest row of the window: the planes moving in sweep direction (no shift, from west, from east)
  float mid[6][4];    // of the middle row: planes 0,1,3 and the three that become `trail`
  float top[9][4];    // newest intermediate row
  uint32_t m_mid, m_top;
  // two row-sets of source loads are kept in flight (ping-pong) so that HBM latency is covered by
  // both collision passes of an iteration
  RowLoads inA, inB;
  issue_row_loads<NTL == 1>(a, wrap(r0), xcol, xhalo_w, xhalo_e, lane, inA);
  issue_row_loads<NTL == 1>(a, wrap(r0 + d), xcol, xhalo_w, xhalo_e, lane, inB);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    trail[0][v] = trail[1][v] = trail[2][v] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) mid[k][v] = 0.f;
  }
  m_mid = 0;
  // one iteration: intermediate row k from `in` (then refill `in` with row k+2), output row k-1
  auto iterate = [&](int k, RowLoads &in) {
    const float t1 = first_step_row(a, in, wrap(r0 + k * d), top);
    m_top = in.m;
    if (owner && k >= 1 && k <= n) sum1 += t1;  // rows k = 0 and n+1 belong to the neighbouring chunks
    if (k + 2 <= n + 1) {
      if (NTL == 2 && k + 2 < n) issue_row_loads<true>(a, wrap(r0 + (k + 2) * d), xcol, xhalo_w, xhalo_e, lane, in);
      else issue_row_loads<NTL == 1>(a, wrap(r0 + (k + 2) * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
    // step t+2 of the middle row of the window (complete from the third iteration on); y-shifted values come
    // from this lane's own registers, x-shifted ones by DPP
    if (k >= 2) {
      const int y = r0 + (k - 1) * d;
      float g[9][4], o[9][4], a1[4], a2[4], b1[4], b2[4], t[4];
      shift_from_west(mid[1], 0.f, g[1]);
      shift_from_east(mid[2], 0.f, g[3]);
      shift_from_west(trail[1], 0.f, a1);
      shift_from_east(trail[2], 0.f, a2);
#pragma unroll
      for (int v = 0; v < 4; v++) t[v] = up ? top[8][v] : top[5][v];
      shift_from_west(t, 0.f, b1);
#pragma unroll
      for (int v = 0; v < 4; v++) t[v] = up ? top[7][v] : top[6][v];
      shift_from_east(t, 0.f, b2);
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const float b0 = up ? top[4][v] : top[2][v];
        g[0][v] = mid[0][v];
        g[2][v] = up ? trail[0][v] : b0;
        g[4][v] = up ? b0 : trail[0][v];
        g[5][v] = up ? a1[v] : b1[v];
        g[8][v] = up ? b1[v] : a1[v];
        g[6][v] = up ? a2[v] : b2[v];
        g[7][v] = up ? b2[v] : a2[v];
      }
      const float t2 = collide4(g, m_mid, a.omega, (y == a.accel_row || y == a.accel_row_b) && a.accel_next, a.aw1, a.aw2, o);
      if (owner) {
        sum2 += t2;
        float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
        for (int kk = 0; kk < 9; kk++) store4<NT>(dp + kk * ps, o[kk][0], o[kk][1], o[kk][2], o[kk][3]);
      }
    }
    // rotate the register window one row in sweep direction
#pragma unroll
    for (int v = 0; v < 4; v++) {
      trail[0][v] = mid[3][v]; trail[1][v] = mid[4][v]; trail[2][v] = mid[5][v];
      mid[0][v] = top[0][v]; mid[1][v] = top[1][v]; mid[2][v] = top[3][v];
      mid[3][v] = up ? top[2][v] : top[4][v];
      mid[4][v] = up ? top[5][v] : top[8][v];
      mid[5][v] = up ? top[6][v] : top[7][v];
    }
    m_mid = m_top;
  };
  for (int k = 0; k <= n + 1; k += 2) {
    iterate(k, inA);
    if (k + 1 <= n + 1) iterate(k + 1, inB);
  }
  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  if (lane == 0) {
    a.partials1[unit] = sum1;
    a.partials2[unit] = sum2;
  }
}

// ---- three timesteps per launch --------------------------------------------------------------------
// The same idea one level deeper: TWO register windows (the state after step t+1 and after step t+2), three
// collision passes per iteration; the grid is read once and written once per THREE steps.  Two halo lanes at
// either end of a wave (the second window is valid on lanes 1..62, the output on lanes 2..61) and two redundant
// intermediate rows at either end of a chunk for the first window, one for the second.  Same helpers, same
// per-cell arithmetic, bit-identical to three single steps.  partials3 receives step t+3's sums.
struct Window {
  float trail[3][4];  // oldest row: the three planes moving in sweep direction (no shift, from west, from east)
  float mid[6][4];    // middle row: planes 0,1,3 and the three that become `trail`
  uint32_t m_mid;     // obstacle flags of the middle row's cells
};

// gather of one step from a window + the newest row `top` (see d2q9_step2), direction-dependent plane roles
__device__ __forceinline__ void window_gather(const Window &w, const float (&top)[9][4], bool up, float (&g)[9][4]) {
  float a1[4], a2[4], b1[4], b2[4], t[4];
  shift_from_west(w.mid[1], 0.f, g[1]);
  shift_from_east(w.mid[2], 0.f, g[3]);
  shift_from_west(w.trail[1], 0.f, a1);
  shift_from_east(w.trail[2], 0.f, a2);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[8][v] : top[5][v];
  shift_from_west(t, 0.f, b1);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[7][v] : top[6][v];
  shift_from_east(t, 0.f, b2);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const float b0 = up ? top[4][v] : top[2][v];
    g[0][v] = w.mid[0][v];
    g[2][v] = up ? w.trail[0][v] : b0;
    g[4][v] = up ? b0 : w.trail[0][v];
    g[5][v] = up ? a1[v] : b1[v];
    g[8][v] = up ? b1[v] : a1[v];
    g[6][v] = up ? a2[v] : b2[v];
    g[7][v] = up ? b2[v] : a2[v];
  }
}

__device__ __forceinline__ void window_rotate(Window &w, const float (&top)[9][4], uint32_t m_top, bool up) {
#pragma unroll
  for (int v = 0; v < 4; v++) {
    w.trail[0][v] = w.mid[3][v]; w.trail[1][v] = w.mid[4][v]; w.trail[2][v] = w.mid[5][v];
    w.mid[0][v] = top[0][v]; w.mid[1][v] = top[1][v]; w.mid[2][v] = top[3][v];
    w.mid[3][v] = up ? top[2][v] : top[4][v];
    w.mid[4][v] = up ? top[5][v] : top[8][v];
    w.mid[5][v] = up ? top[6][v] : top[7][v];
  }
  w.m_mid = m_top;
}

// LDS-resident form of a Window: a workgroup is ONE wave, so its LDS is private to the wave and needs no barrier
// (a wave's LDS operations complete in order).  Slot s of a window holds one float4 per lane at (s*64 + lane);
// slots 0..2 = planes 0,1,3 of the middle row, slots 3+3p .. 5+3p = the sweep-direction planes of the rows of
// parity p (written in iteration k, read as `trail` in iteration k+2).  Moving both windows (72 registers) to
// LDS brings the kernel under 256 registers: two waves per SIMD instead of one.
constexpr int kWinSlots = 9;
__device__ __forceinline__ void lds_put(v4f *w, int slot, const float (&p)[4]) {
  v4f v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
  w[slot * 64] = v;
}
__device__ __forceinline__ void lds_get(const v4f *w, int slot, float (&p)[4]) {
  const v4f v = w[slot * 64];
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

// gather from an LDS window (w already offset by the lane) + the newest row `top`, then store `top` as the
// window's new middle row: the LDS counterpart of window_gather followed by window_rotate
__device__ __forceinline__ void lds_window_gather(const v4f *w, int par, const float (&top)[9][4], bool up, float (&g)[9][4]) {
  float a1[4], a2[4], b1[4], b2[4], t[4], tr0[4];
  lds_get(w, 1, t); shift_from_west(t, 0.f, g[1]);
  lds_get(w, 2, t); shift_from_east(t, 0.f, g[3]);
  lds_get(w, 0, g[0]);
  lds_get(w, 3 + 3 * par, tr0);
  lds_get(w, 4 + 3 * par, t); shift_from_west(t, 0.f, a1);
  lds_get(w, 5 + 3 * par, t); shift_from_east(t, 0.f, a2);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[8][v] : top[5][v];
  shift_from_west(t, 0.f, b1);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[7][v] : top[6][v];
  shift_from_east(t, 0.f, b2);
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const float b0 = up ? top[4][v] : top[2][v];
    g[2][v] = up ? tr0[v] : b0;
    g[4][v] = up ? b0 : tr0[v];
    g[5][v] = up ? a1[v] : b1[v];
    g[8][v] = up ? b1[v] : a1[v];
    g[6][v] = up ? a2[v] : b2[v];
    g[7][v] = up ? b2[v] : a2[v];
  }
}

__device__ __forceinline__ void lds_window_put(v4f *w, int par, const float (&top)[9][4], bool up) {
  float t[4];
  lds_put(w, 0, top[0]); lds_put(w, 1, top[1]); lds_put(w, 2, top[3]);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[2][v] : top[4][v];
  lds_put(w, 3 + 3 * par, t);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[5][v] : top[8][v];
  lds_put(w, 4 + 3 * par, t);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[6][v] : top[7][v];
  lds_put(w, 5 + 3 * par, t);
}

// WLDS: the two windows live in LDS (two waves per SIMD) instead of registers (one wave per SIMD).
// NBUF: row-sets of source loads in flight (2 = ping-pong as in d2q9_step2, 1 = the next row only).
template <bool NT, int NTL = 0, bool WLDS = false, int NBUF = 2, bool PUSH = false>
__global__ __launch_bounds__(64) void d2q9_step3(const Step2Args a, float *partials3) {
  __shared__ v4f win[WLDS ? 2 * kWinSlots * 64 : 1];
  const int lane = threadIdx.x;
  const UnitSel us = select_unit<PUSH>(a, a.edge_units);
  // decided once (re-reading it from the argument block in every row iteration drained the wave's loads each time:
  // 8192x4096 slab 246 instead of 281 GLUPS)
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if ((threadIdx.x & 63) < 2) spin_on_flag(pp->wait_flags + (threadIdx.x & 63), la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;
  const int unit0 = band * us.units_per_band + slot;
  const int chunk = unit0 / a.strips, strip = unit0 - chunk * a.strips;
  const int unit = unit0 + us.partial_off;  // slot of the velocity sums
  const int ys = us.chunk_start[chunk];
  const int ye = us.chunk_start[chunk + 1];
  if (ys >= ye || chunk == us.skip) {
    if (lane == 0) a.partials1[unit] = a.partials2[unit] = partials3[unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;
  }
  const int q4 = a.nx >> 2;
  const int qcol = strip * a.lanes_out + lane - 2;  // lanes 0,1 and 62,63 are halo lanes
  const bool owner = (lane >= 2) && (lane < 2 + a.lanes_out) && (qcol < q4);
  int qw = qcol % q4;
  if (qw < 0) qw += q4;
  const int xcol = qw * 4;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 4 >= a.nx) ? 0 : xcol + 4;
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };

  const bool up = __builtin_amdgcn_readfirstlane((int)((chunk & 1) == 0)) != 0;
  const int n = ye - ys;
  const int d = up ? 1 : -1;
  const int r0 = up ? ys - 2 : ye + 1;  // row of the k-th first-level row: r0 + k*d, k = 0 .. n+3

  float sum1 = 0.f, sum2 = 0.f, sum3 = 0.f;
  Window w1, w2;
  v4f *const lw1 = win + lane, *const lw2 = win + (WLDS ? kWinSlots * 64 : 0) + lane;
  uint32_t m_mid1 = 0, m_mid2 = 0;
  float top1[9][4], top2[9][4];
  RowLoads inA, inB;
  issue_row_loads<NTL == 1>(a, wrap(r0), xcol, xhalo_w, xhalo_e, lane, inA);
  if (NBUF == 2) issue_row_loads<NTL == 1>(a, wrap(r0 + d), xcol, xhalo_w, xhalo_e, lane, inB);
  if (!WLDS) {
#pragma unroll
    for (int v = 0; v < 4; v++) {
#pragma unroll
      for (int k = 0; k < 3; k++) w1.trail[k][v] = w2.trail[k][v] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) w1.mid[k][v] = w2.mid[k][v] = 0.f;
    }
  }

  auto iterate = [&](int k, RowLoads &in, const bool up) __attribute__((always_inline)) {
    const int par = k & 1;
    // level 1: state after step t+1 on row r0 + k*d (rows k = 2 .. n+1 are this chunk's own)
    const int row1 = wrap(r0 + k * d);
    const float t1 = first_step_row(a, in, row1, top1);
    const uint32_t m1 = in.m;
    if (owner && k >= 2 && k <= n + 1) sum1 += t1;
    const int kn = k + NBUF;
    if (kn <= n + 3) {
      // source rows shared with the neighbouring chunk (its own sweep reads them too) stay cacheable
      if (NTL == 2 && kn >= 4 && kn <= n - 1) issue_row_loads<true>(a, wrap(r0 + kn * d), xcol, xhalo_w, xhalo_e, lane, in);
      else issue_row_loads<NTL == 1>(a, wrap(r0 + kn * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
    // level 2: state after step t+2 on the middle row of window 1 (row r0 + (k-1)*d), from the third iteration on
    uint32_t m2 = 0;
    if (k >= 2) {
      const int row2 = wrap(r0 + (k - 1) * d);
      float g[9][4];
      if (WLDS) lds_window_gather(lw1, par, top1, up, g);
      else window_gather(w1, top1, up, g);
      m2 = m_mid1;
      const float t2 = collide4(g, m2, a.omega, row2 == a.accel_row || row2 == a.accel_row_b, a.aw1, a.aw2, top2);
      if (owner && k >= 3 && k <= n + 2) sum2 += t2;
    }
    if (WLDS) lds_window_put(lw1, par, top1, up);
    else window_rotate(w1, top1, m1, up);
    m_mid1 = m1;
    // level 3: step t+3 on the middle row of window 2 (row r0 + (k-2)*d), from the fifth iteration on
    if (k >= 4) {
      const int y = r0 + (k - 2) * d;
      float g[9][4], o[9][4];
      if (WLDS) lds_window_gather(lw2, par, top2, up, g);
      else window_gather(w2, top2, up, g);
      const float t3 = collide4(g, m_mid2, a.omega, (y == a.accel_row || y == a.accel_row_b) && a.accel_next, a.aw1, a.aw2, o);
      if (owner) {
        sum3 += t3;
        float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
        for (int kk = 0; kk < 9; kk++) store4<NT>(dp + kk * ps, o[kk][0], o[kk][1], o[kk][2], o[kk][3]);
      }
    }
    if (k >= 2) {
      if (WLDS) lds_window_put(lw2, par, top2, up);
      else window_rotate(w2, top2, m2, up);
      m_mid2 = m2;
    }
  };
  if (NBUF == 2) {
    for (int k = 0; k <= n + 3; k += 2) {
      iterate(k, inA, up);
      if (k + 1 <= n + 3) iterate(k + 1, inB, up);
    }
  } else {
    // (one copy of the loop per sweep direction, which folds the plane-role selects away, was 2 % slower:
    // twice the code for the instruction cache)
    for (int k = 0; k <= n + 3; k++) iterate(k, inA, up);
  }
  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  sum3 = wave_sum(sum3);
  if (lane == 0) {
    a.partials1[unit] = sum1;
    a.partials2[unit] = sum2;
    partials3[unit] = sum3;
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      push_chunk(la, ys, ye, xcol, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// ---- three timesteps per launch, chunk pairs ------------------------------------------------------------
// d2q9_step3 (LDS windows, one row-set of loads in flight) with the start-up redundancy removed: a workgroup is
// TWO waves, the chunks 2p (sweeping down) and 2p+1 (sweeping up) of one strip, which START at their common
// boundary at the same time.  What a lone wave computes redundantly to prime its windows — two first-level rows
// and one second-level row beyond its start, the partner's first rows — the partner computes anyway: each wave
// writes the three planes of its first first-level row (iteration 2) and of its first second-level row
// (iteration 3) that move towards the partner straight into the `trail` slots of the partner's LDS windows,
// where the partner's ordinary gather finds them one iteration later.  Two barriers per workgroup, no extra LDS,
// n+2 iterations and 3n+3 collision passes per chunk of n rows instead of n+4 and 3n+6.  A wave whose partner
// is empty or skipped runs alone exactly like d2q9_step3.  Same arithmetic, bit-identical.
__device__ __forceinline__ void lds_put_trail(v4f *w, int par, const float (&top)[9][4], bool up) {
  float t[4];
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[2][v] : top[4][v];
  lds_put(w, 3 + 3 * par, t);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[5][v] : top[8][v];
  lds_put(w, 4 + 3 * par, t);
#pragma unroll
  for (int v = 0; v < 4; v++) t[v] = up ? top[6][v] : top[7][v];
  lds_put(w, 5 + 3 * par, t);
}

template <bool NT, int NTL = 0, bool PUSH = false>
__global__ __launch_bounds__(128) void d2q9_step3p(const Step2Args a, float *partials3) {
  __shared__ v4f win[2 * 2 * kWinSlots * 64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const UnitSel us = select_unit<PUSH>(a, a.edge_units / 2);
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if ((threadIdx.x & 63) < 2) spin_on_flag(pp->wait_flags + (threadIdx.x & 63), la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;  // units_per_band counts chunk PAIRS x strips here
  const int punit = band * us.units_per_band + slot;
  const int pair = punit / a.strips, strip = punit - pair * a.strips;
  const int chunk = 2 * pair + wv;
  const int unit = chunk * a.strips + strip + us.partial_off;  // slot of the partial sums, as in d2q9_step3
  const int ys = us.chunk_start[chunk];
  const int ye = us.chunk_start[chunk + 1];
  const int pys = us.chunk_start[chunk ^ 1], pye = us.chunk_start[(chunk ^ 1) + 1];
  const bool empty = ys >= ye || chunk == us.skip;
  const bool paired = !empty && pys < pye && (chunk ^ 1) != us.skip;  // the same on both waves
  if (empty) {
    if (lane == 0) a.partials1[unit] = a.partials2[unit] = partials3[unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;  // the partner then runs alone and meets no barrier
  }
  const int q4 = a.nx >> 2;
  const int qcol = strip * a.lanes_out + lane - 2;
  const bool owner = (lane >= 2) && (lane < 2 + a.lanes_out) && (qcol < q4);
  int qw = qcol % q4;
  if (qw < 0) qw += q4;
  const int xcol = qw * 4;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 4 >= a.nx) ? 0 : xcol + 4;
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };

  const bool up = wv != 0;  // even chunks sweep down from their top, odd ones up from their bottom: pairs start together
  const int n = ye - ys;
  const int d = up ? 1 : -1;
  const int r0 = up ? ys - 2 : ye + 1;
  const int kbeg = paired ? 2 : 0;   // the first own row is k = 2
  const int k2 = paired ? 3 : 2;     // first iteration with a second-level row

  float sum1 = 0.f, sum2 = 0.f, sum3 = 0.f;
  v4f *const mine = win + wv * (2 * kWinSlots * 64) + lane, *const theirs = win + (wv ^ 1) * (2 * kWinSlots * 64) + lane;
  v4f *const lw1 = mine, *const lw2 = mine + kWinSlots * 64;
  uint32_t m_mid1 = 0, m_mid2 = 0;
  float top1[9][4], top2[9][4];
  RowLoads in;
  issue_row_loads<NTL == 1>(a, wrap(r0 + kbeg * d), xcol, xhalo_w, xhalo_e, lane, in);
  for (int k = kbeg; k <= n + 3; k++) {
    const int par = k & 1;
    const int row1 = wrap(r0 + k * d);
    const float t1 = first_step_row(a, in, row1, top1);
    const uint32_t m1 = in.m;
    if (owner && k >= 2 && k <= n + 1) sum1 += t1;
    if (k + 1 <= n + 3) {
      if (NTL == 2 && k + 1 >= 4 && k + 1 <= n - 1) issue_row_loads<true>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
      else issue_row_loads<NTL == 1>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
    uint32_t m2 = 0;
    if (k >= k2) {
      const int row2 = wrap(r0 + (k - 1) * d);
      float g[9][4];
      lds_window_gather(lw1, par, top1, up, g);
      m2 = m_mid1;
      const float t2 = collide4(g, m2, a.omega, row2 == a.accel_row || row2 == a.accel_row_b, a.aw1, a.aw2, top2);
      if (owner && k >= 3 && k <= n + 2) sum2 += t2;
    }
    lds_window_put(lw1, par, top1, up);
    m_mid1 = m1;
    if (paired && k == 2) {
      // my first first-level row is the row just across the partner's start: its planes moving the partner's way are
      // what the partner's gather of iteration 3 expects in the trail slots of its window 1 (parity 1)
      lds_put_trail(theirs, 1, top1, !up);
      __syncthreads();
    }
    if (k >= 4) {
      const int y = r0 + (k - 2) * d;
      float g[9][4], o[9][4];
      lds_window_gather(lw2, par, top2, up, g);
      const float t3 = collide4(g, m_mid2, a.omega, (y == a.accel_row || y == a.accel_row_b) && a.accel_next, a.aw1, a.aw2, o);
      if (owner) {
        sum3 += t3;
        float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
        for (int kk = 0; kk < 9; kk++) store4<NT>(dp + kk * ps, o[kk][0], o[kk][1], o[kk][2], o[kk][3]);
      }
    }
    if (k >= k2) {
      lds_window_put(lw2, par, top2, up);
      m_mid2 = m2;
    }
    if (paired && k == 3) {
      // the same one level up: my first second-level row into the trail slots (parity 0) of the partner's window 2
      lds_put_trail(theirs + kWinSlots * 64, 0, top2, !up);
      __syncthreads();
    }
  }
  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  sum3 = wave_sum(sum3);
  if (lane == 0) {
    a.partials1[unit] = sum1;
    a.partials2[unit] = sum2;
    partials3[unit] = sum3;
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      push_chunk(la, ys, ye, xcol, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// ---- four timesteps per launch ------------------------------------------------------------------------
// d2q9_step3 with one more level: windows 1 and 2 in LDS as before, window 3 in the registers that kernel leaves free.  Level-1 rows k = 0 .. n+5 (row r0 + k*d), level-2 row
// k-1 from k = 2, level-3 row k-2 from k = 4, output row k-3 from k = 6.  One cell of the outermost lane becomes
// invalid per level, so the two halo lanes per side of d2q9_step3 are enough here too (60 output lanes).
// Two planes of window 3 go to the 2 KB of LDS per wave that two waves per SIMD leave free (8 x 20 KB = the CU's
// 160 KB), and the row loads use SGPR-base addressing: together they bring the kernel from 254 VGPRs + 12 spilled
// to 253 and none.  The spills cost 17 % (8192x8192: 249 -> 274 GLUPS with the LDS planes, -> 291 with both).
constexpr int kW3Lds = 2;
template <bool NT, int NTL = 0, bool PUSH = false>
__global__ __launch_bounds__(64, 2) void d2q9_step4(const Step2Args a, float *partials3, float *partials4) {
  __shared__ v4f win[(2 * kWinSlots + kW3Lds) * 64];
  const int lane = threadIdx.x;
  const UnitSel us = select_unit<PUSH>(a, a.edge_units);
  // decided once (re-reading it from the argument block in every row iteration drained the wave's loads each time:
  // 8192x4096 slab 246 instead of 281 GLUPS)
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if ((threadIdx.x & 63) < 2) spin_on_flag(pp->wait_flags + (threadIdx.x & 63), la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;
  const int unit0 = band * us.units_per_band + slot;
  const int chunk = unit0 / a.strips, strip = unit0 - chunk * a.strips;
  const int unit = unit0 + us.partial_off;  // slot of the velocity sums
  const int ys = us.chunk_start[chunk];
  const int ye = us.chunk_start[chunk + 1];
  if (ys >= ye || chunk == us.skip) {
    if (lane == 0) a.partials1[unit] = a.partials2[unit] = partials3[unit] = partials4[unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;
  }
  const int q4 = a.nx >> 2;
  const int qcol = strip * a.lanes_out + lane - 2;
  const bool owner = (lane >= 2) && (lane < 2 + a.lanes_out) && (qcol < q4);
  int qw = qcol % q4;
  if (qw < 0) qw += q4;
  const int xcol = qw * 4;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 4 >= a.nx) ? 0 : xcol + 4;
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };
  const bool up = __builtin_amdgcn_readfirstlane((int)((chunk & 1) == 0)) != 0;
  const int n = ye - ys;
  const int d = up ? 1 : -1;
  const int r0 = up ? ys - 3 : ye + 2;

  float sum1 = 0.f, sum2 = 0.f, sum3 = 0.f, sum4 = 0.f;
  v4f *const lw1 = win + lane, *const lw2 = win + kWinSlots * 64 + lane, *const lw3 = win + 2 * kWinSlots * 64 + lane;
  Window w3;
  uint32_t m_mid1 = 0, m_mid2 = 0, m_mid3 = 0;
  float top1[9][4], top2[9][4], top3[9][4];
  RowLoads in;
  issue_row_loads_sbase<NTL == 1>(a, wrap(r0), xcol, xhalo_w, xhalo_e, lane, in);
#pragma unroll
  for (int v = 0; v < 4; v++) {
#pragma unroll
    for (int k = 0; k < 3; k++) w3.trail[k][v] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) w3.mid[k][v] = 0.f;
  }
  for (int k = 0; k <= n + 5; k++) {
    const int par = k & 1;
    const int row1 = wrap(r0 + k * d);
    const float t1 = first_step_row(a, in, row1, top1);
    const uint32_t m1 = in.m;
    if (owner && k >= 3 && k <= n + 2) sum1 += t1;
    if (k + 1 <= n + 5) {
      if (NTL == 2 && k + 1 >= 6 && k + 1 <= n - 1) issue_row_loads_sbase<true>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
      else issue_row_loads_sbase<NTL == 1>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
    uint32_t m2 = 0, m3 = 0;
    if (k >= 2) {
      const int row2 = wrap(r0 + (k - 1) * d);
      float g[9][4];
      lds_window_gather(lw1, par, top1, up, g);
      m2 = m_mid1;
      const float t2 = collide4(g, m2, a.omega, row2 == a.accel_row || row2 == a.accel_row_b, a.aw1, a.aw2, top2);
      if (owner && k >= 4 && k <= n + 3) sum2 += t2;
    }
    lds_window_put(lw1, par, top1, up);
    m_mid1 = m1;
    if (k >= 4) {
      const int row3 = wrap(r0 + (k - 2) * d);
      float g[9][4];
      lds_window_gather(lw2, par, top2, up, g);
      m3 = m_mid2;
      const float t3 = collide4(g, m3, a.omega, row3 == a.accel_row || row3 == a.accel_row_b, a.aw1, a.aw2, top3);
      if (owner && k >= 5 && k <= n + 4) sum3 += t3;
    }
    if (k >= 2) {
      lds_window_put(lw2, par, top2, up);
      m_mid2 = m2;
    }
    if (k >= 6) {
      const int y = r0 + (k - 3) * d;
      float g[9][4], o[9][4];
      if (kW3Lds == 2) { lds_get(lw3, 0, w3.mid[0]); lds_get(lw3, 1, w3.mid[1]); }
      window_gather(w3, top3, up, g);
      const float t4 = collide4(g, m_mid3, a.omega, (y == a.accel_row || y == a.accel_row_b) && a.accel_next, a.aw1, a.aw2, o);
      if (owner) {
        sum4 += t4;
        float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
        for (int kk = 0; kk < 9; kk++) store4<NT>(dp + kk * ps, o[kk][0], o[kk][1], o[kk][2], o[kk][3]);
      }
    }
    if (k >= 4) {
      window_rotate(w3, top3, m3, up);
      if (kW3Lds == 2) { lds_put(lw3, 0, w3.mid[0]); lds_put(lw3, 1, w3.mid[1]); }
      m_mid3 = m3;
    }
  }
  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  sum3 = wave_sum(sum3);
  sum4 = wave_sum(sum4);
  if (lane == 0) {
    a.partials1[unit] = sum1;
    a.partials2[unit] = sum2;
    partials3[unit] = sum3;
    partials4[unit] = sum4;
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      push_chunk(la, ys, ye, xcol, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// ---- four timesteps per launch, chunk pairs -------------------------------------------------------------
// d2q9_step4 with the start-up redundancy removed the way d2q9_step3p does it: the chunks 2p (down) and 2p+1 (up) of a
// strip start at their common boundary as one workgroup of two waves and hand each other their first row of every
// level — levels 1 and 2 into the trail slots of the partner's LDS windows (iterations 3 and 4), level 3 through
// the partner's window-2 trail slots once its own level-3 gather has consumed them (iteration 5, two barriers) and
// from there into the register window.  n+3 iterations and 4n+6 collision passes per chunk instead of n+6 and 4n+12.
template <bool NT, int NTL = 0, bool PUSH = false>
__global__ __launch_bounds__(128, 2) void d2q9_step4p(const Step2Args a, float *partials3, float *partials4) {
  constexpr int kWave = (2 * kWinSlots + kW3Lds) * 64;  // LDS float4s per wave
  __shared__ v4f win[2 * kWave];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const UnitSel us = select_unit<PUSH>(a, a.edge_units / 2);
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if ((threadIdx.x & 63) < 2) spin_on_flag(pp->wait_flags + (threadIdx.x & 63), la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;  // units_per_band counts chunk PAIRS x strips here
  const int punit = band * us.units_per_band + slot;
  const int pair = punit / a.strips, strip = punit - pair * a.strips;
  const int chunk = 2 * pair + wv;
  const int unit = chunk * a.strips + strip + us.partial_off;
  const int ys = us.chunk_start[chunk];
  const int ye = us.chunk_start[chunk + 1];
  const int pys = us.chunk_start[chunk ^ 1], pye = us.chunk_start[(chunk ^ 1) + 1];
  const bool empty = ys >= ye || chunk == us.skip;
  const bool paired = !empty && pys < pye && (chunk ^ 1) != us.skip;  // the same on both waves
  if (empty) {
    if (lane == 0) a.partials1[unit] = a.partials2[unit] = partials3[unit] = partials4[unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;
  }
  const int q4 = a.nx >> 2;
  const int qcol = strip * a.lanes_out + lane - 2;
  const bool owner = (lane >= 2) && (lane < 2 + a.lanes_out) && (qcol < q4);
  int qw = qcol % q4;
  if (qw < 0) qw += q4;
  const int xcol = qw * 4;
  // (ONE register for both: only lane 0 reads the element west of its strip, only lane 63 the one east of it — the
  // compact-set form of this kernel sat two registers over the limit, 12 bytes of scratch per lane)
  const int xhalo_w = (lane == 0) ? ((xcol == 0) ? a.nx - 1 : xcol - 1) : ((xcol + 4 >= a.nx) ? 0 : xcol + 4);
  const int xhalo_e = xhalo_w;
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };
  const bool up = wv != 0;
  const int n = ye - ys;
  const int d = up ? 1 : -1;
  const int r0 = up ? ys - 3 : ye + 2;

  float sum1 = 0.f, sum2 = 0.f, sum3 = 0.f, sum4 = 0.f;
  v4f *const mine = win + wv * kWave + lane, *const theirs = win + (wv ^ 1) * kWave + lane;
  v4f *const lw1 = mine, *const lw2 = mine + kWinSlots * 64, *const lw3 = mine + 2 * kWinSlots * 64;
  Window w3;
  uint32_t m_mid1 = 0, m_mid2 = 0, m_mid3 = 0;
  float top1[9][4], top2[9][4], top3[9][4];
  RowLoads in;
  issue_row_loads_sbase<NTL == 1>(a, wrap(r0 + (paired ? 3 : 0) * d), xcol, xhalo_w, xhalo_e, lane, in);
#pragma unroll
  for (int v = 0; v < 4; v++) {
#pragma unroll
    for (int k = 0; k < 3; k++) w3.trail[k][v] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) w3.mid[k][v] = 0.f;
  }
  // One iteration.  PHASE 0 = the steady loop (and the whole sweep of a wave without partner: levels switch on at
  // k = 2, 4, 6); PHASE 3, 4, 5 = the three start-up iterations of a pair, peeled so that their hand-overs do not
  // lengthen any live range inside the steady loop (the kernel has no register to spare).
  auto iter = [&](const int k, const int PHASE) __attribute__((always_inline)) {
    const int par = k & 1;
    const int row1 = wrap(r0 + k * d);
    const float t1 = first_step_row(a, in, row1, top1);
    const uint32_t m1 = in.m;
    if (owner && k >= 3 && k <= n + 2) sum1 += t1;
    if (k + 1 <= n + 5) {
      if (NTL == 2 && k + 1 >= 6 && k + 1 <= n - 1) issue_row_loads_sbase<true>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
      else issue_row_loads_sbase<NTL == 1>(a, wrap(r0 + (k + 1) * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
    uint32_t m2 = 0, m3 = 0;
    if (PHASE == 0 ? k >= 2 : PHASE >= 4) {
      const int row2 = wrap(r0 + (k - 1) * d);
      float g[9][4];
      lds_window_gather(lw1, par, top1, up, g);
      m2 = m_mid1;
      const float t2 = collide4(g, m2, a.omega, row2 == a.accel_row || row2 == a.accel_row_b, a.aw1, a.aw2, top2);
      if (owner && k >= 4 && k <= n + 3) sum2 += t2;
    }
    lds_window_put(lw1, par, top1, up);
    m_mid1 = m1;
    if (PHASE == 3) {
      lds_put_trail(theirs, 0, top1, !up);  // my first level-1 row: the partner's trail at its iteration 4
      __syncthreads();
    }
    if (PHASE == 0 ? k >= 4 : PHASE == 5) {
      const int row3 = wrap(r0 + (k - 2) * d);
      float g[9][4];
      lds_window_gather(lw2, par, top2, up, g);
      m3 = m_mid2;
      const float t3 = collide4(g, m3, a.omega, row3 == a.accel_row || row3 == a.accel_row_b, a.aw1, a.aw2, top3);
      if (owner && k >= 5 && k <= n + 4) sum3 += t3;
    }
    if (PHASE == 5) {
      // my first level-3 row goes through the trail slots of the partner's window 2 (parity 1), which its level-3
      // gather of this iteration has just consumed, into the partner's register window
      __syncthreads();
      lds_put_trail(theirs + kWinSlots * 64, 1, top3, !up);
      __syncthreads();
      window_rotate(w3, top3, m3, up);
      lds_get(lw2, 3 + 3, w3.trail[0]); lds_get(lw2, 4 + 3, w3.trail[1]); lds_get(lw2, 5 + 3, w3.trail[2]);
      if (kW3Lds == 2) { lds_put(lw3, 0, w3.mid[0]); lds_put(lw3, 1, w3.mid[1]); }
      m_mid3 = m3;
    }
    if (PHASE == 0 ? k >= 2 : PHASE >= 4) {
      lds_window_put(lw2, par, top2, up);
      m_mid2 = m2;
    }
    if (PHASE == 4) {
      lds_put_trail(theirs + kWinSlots * 64, 1, top2, !up);  // my first level-2 row: the partner's trail at its iteration 5
      __syncthreads();
    }
    if (PHASE == 0) {
      if (k >= 6) {
        const int y = r0 + (k - 3) * d;
        float g[9][4], o[9][4];
        if (kW3Lds == 2) { lds_get(lw3, 0, w3.mid[0]); lds_get(lw3, 1, w3.mid[1]); }
        window_gather(w3, top3, up, g);
        const float t4 = collide4(g, m_mid3, a.omega, (y == a.accel_row || y == a.accel_row_b) && a.accel_next, a.aw1, a.aw2, o);
        if (owner) {
          sum4 += t4;
          float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
          for (int kk = 0; kk < 9; kk++) store4<NT>(dp + kk * ps, o[kk][0], o[kk][1], o[kk][2], o[kk][3]);
        }
      }
      if (k >= 4) {
        window_rotate(w3, top3, m3, up);
        if (kW3Lds == 2) { lds_put(lw3, 0, w3.mid[0]); lds_put(lw3, 1, w3.mid[1]); }
        m_mid3 = m3;
      }
    }
  };
  int kstart = 0;
  if (paired) {
    iter(3, 3);
    iter(4, 4);
    iter(5, 5);
    kstart = 6;
  }
  for (int k = kstart; k <= n + 5; k++) iter(k, 0);
  sum1 = wave_sum(sum1);
  sum2 = wave_sum(sum2);
  sum3 = wave_sum(sum3);
  sum4 = wave_sum(sum4);
  if (lane == 0) {
    a.partials1[unit] = sum1;
    a.partials2[unit] = sum2;
    partials3[unit] = sum3;
    partials4[unit] = sum4;
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      // the lane's column once more, from the lane number (hidden from common-subexpression elimination): kept from the
      // top of the kernel its 64-bit form was live across the row loop — the two registers this kernel spilled
      int lane2 = (int)(threadIdx.x & 63);
      asm volatile("" : "+v"(lane2));
      int qw2 = (strip * la->lanes_out + lane2 - 2) % q4;
      if (qw2 < 0) qw2 += q4;
      push_chunk(la, ys, ye, qw2 * 4, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// ---- D timesteps per launch, two cells per lane ("deep" window kernel) --------------------------------------
// The window kernels above hold four cells per lane: two LDS windows and one register window are all that fits two
// waves per SIMD, so four steps per launch is their limit, and at four steps they move real traffic at 90-93 % of what
// a copy reaches — only fewer bytes per update make the big grids faster.  This kernel halves the per-wave state
// instead: a lane holds TWO neighbouring cells (one v2f per plane), a window slot is 512 B, D-1 windows fit
// (D = 8: four in LDS, 18 KB per wave, three in registers), the grid is read once and written once per D steps.
//  - the collision is written on explicit pairs (collide_pair: the operations of collide_cell, in its order, on both
//    cells at once) and compiles to v_pk_* instructions without the pack/unpack moves the float4 kernels need;
//  - planes that move along x are read back from the LDS window already shifted by one cell (an unaligned 8-byte read
//    at +-4 bytes: ds_read2_b32) instead of DPP + moves; a wave runs ONE sweep direction as a template parameter, so
//    the plane roles are fixed at compile time (no selects) — per cell and step 64 VALU instructions instead of 92;
//  - the depth is a launch argument (nlev <= D): a run is cut into the fewest launches, of equal depth; for the depths runs
//    are actually cut into (6, 7, 8; the twins' 5) there is a kernel instantiated per depth (template value LT) whose row
//    loop switches, once all levels are running, to a STEADY form with a straight-line level chain (see deep_sweep);
//  - the last level's row is stored one iteration late, right before that iteration's loads (see deep_sweep);
//  - what is known about a row (is it the accelerated row, is it one of the chunk's own rows) is computed once, for
//    level 0, and travels to the deeper levels in scalar shift registers: level l works on the row level 0 had l
//    iterations earlier.
// Level l's output is valid from cell l inwards at either end of a strip, so a strip keeps 64 - 2*ceil((D-1)/2) output
// lanes (56 for D = 8).  Same arithmetic per cell as everything else: bit-identical to D single steps.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { v2f r = {x, x}; return r; }

// collide_cell on the pair (cell a, cell b): every statement is collide_cell's, applied to both components; the sum
// order of the density, the grouping of the momenta and every fused multiply-add are the same, so each component
// rounds exactly as collide_cell does.  MAY_BE_OBSTACLE = false: the caller knows that no cell of the WAVE is blocked.
template <bool MAY_BE_OBSTACLE = true>
__device__ __forceinline__ v2f collide_pair(const v2f (&g)[9], bool oa_in, bool ob_in, float omega, v2f (&out)[9]) {
#pragma clang fp contract(off)
  const bool oa = MAY_BE_OBSTACLE && oa_in, ob = MAY_BE_OBSTACLE && ob_in;
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;
  v2f dens = g[0] + g[1];
  dens += g[2]; dens += g[3]; dens += g[4]; dens += g[5]; dens += g[6]; dens += g[7]; dens += g[8];
  const v2f densinv = {__builtin_amdgcn_rcpf(dens.x), __builtin_amdgcn_rcpf(dens.y)};
  const v2f da = g[5] - g[7], db = g[8] - g[6];
  const v2f jx = (g[1] - g[3]) + (da + db);
  const v2f jy = (g[2] - g[4]) + (da - db);
  const v2f usq = fma2(jx, jx, jy * jy);
  const v2f h = splat2(1.5f) * densinv;
  const v2f c = fma2(-h, usq, dens);
  const v2f h3 = splat2(3.0f) * h;
  const v2f jp = jx + jy, jm = jx - jy;
  const v2f ax = fma2(h3 * jx, jx, c), ay = fma2(h3 * jy, jy, c), ap = fma2(h3 * jp, jp, c), am = fma2(h3 * jm, jm, c);
  v2f eq[9];
  eq[0] = splat2(w0) * c;
  eq[1] = splat2(w1) * fma2(splat2(3.0f), jx, ax); eq[3] = splat2(w1) * fma2(splat2(-3.0f), jx, ax);
  eq[2] = splat2(w1) * fma2(splat2(3.0f), jy, ay); eq[4] = splat2(w1) * fma2(splat2(-3.0f), jy, ay);
  eq[5] = splat2(w2) * fma2(splat2(3.0f), jp, ap); eq[7] = splat2(w2) * fma2(splat2(-3.0f), jp, ap);
  eq[8] = splat2(w2) * fma2(splat2(3.0f), jm, am); eq[6] = splat2(w2) * fma2(splat2(-3.0f), jm, am);
  constexpr int opp[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};  // kernels.cl:69
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const v2f r = fma2(splat2(omega), eq[k] - g[k], g[k]);
    out[k].x = oa ? g[opp[k]].x : r.x;
    out[k].y = ob ? g[opp[k]].y : r.y;
  }
  const v2f root = {__builtin_amdgcn_sqrtf(usq.x), __builtin_amdgcn_sqrtf(usq.y)};
  v2f u = root * densinv;
  u.x = oa ? 0.f : u.x;
  u.y = ob ? 0.f : u.y;
  return u;
}

// accelerate_cell on both cells of a pair (kernels.cl:24-42)
__device__ __forceinline__ void accelerate_pair(v2f (&f)[9], bool oa, bool ob, float aw1, float aw2) {
#pragma clang fp contract(off)
  const bool ta = !oa && (f[3].x - aw1) > 0.0f && (f[6].x - aw2) > 0.0f && (f[7].x - aw2) > 0.0f;
  const bool tb = !ob && (f[3].y - aw1) > 0.0f && (f[6].y - aw2) > 0.0f && (f[7].y - aw2) > 0.0f;
  if (ta) { f[1].x += aw1; f[5].x += aw2; f[8].x += aw2; f[3].x -= aw1; f[6].x -= aw2; f[7].x -= aw2; }
  if (tb) { f[1].y += aw1; f[5].y += aw2; f[8].y += aw2; f[3].y -= aw1; f[6].y -= aw2; f[7].y -= aw2; }
}

// one step on a pair: collision, then the next step's accelerate_flow if this is the accelerated row (the row is the
// same for the whole wave: a scalar branch).  OBST = false: no cell of the wave is blocked.
template <bool OBST>
__device__ __forceinline__ v2f collide2(const v2f (&g)[9], uint32_t m, float omega, bool accel_uniform, float aw1, float aw2,
                                        v2f (&o)[9]) {
  const bool oa = (m & 0xffu) != 0, ob = (m & 0xff00u) != 0;
  const v2f u = collide_pair<OBST>(g, oa, ob, omega, o);
  if (accel_uniform) accelerate_pair(o, OBST && oa, OBST && ob, aw1, aw2);
  return u;  // |j|/rho of the two cells (0 for a blocked one): summed per component, the components added at the end
}

// the pair shifted by one cell: lane i receives the odd cell of lane i-1 / the even cell of lane i+1
__device__ __forceinline__ v2f pair_from_west(v2f p, float halo) { v2f r = {dpp_from_lane_below(p.y, halo), p.x}; return r; }
__device__ __forceinline__ v2f pair_from_east(v2f p, float halo) { v2f r = {p.y, dpp_from_lane_above(p.x, halo)}; return r; }
__device__ __forceinline__ v2f pair_from_west(v2f p) {
  v2f r = {__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p.y), 0x138, 0xf, 0xf, true)), p.x};
  return r;
}
__device__ __forceinline__ v2f pair_from_east(v2f p) {
  v2f r = {p.y, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p.x), 0x130, 0xf, 0xf, true))};
  return r;
}

struct PairLoads {
  v2f c[9];
  float h0, h1, h2;  // lane 0: west neighbours of planes 1,5,8; lane 63: east neighbours of planes 3,6,7
  uint32_t m;        // the aligned 4 mask bytes that hold the two cells' (bytes 0,1 or 2,3: see deep_sweep's mask_shift)
};
// HALO = false: no neighbour elements for lanes 0 and 63 (the strip's halo lanes cover the depth without them, see
// deep_edge_loads)
template <bool NTL, bool HALO = true>
__device__ __forceinline__ void issue_pair_loads(const Step2Args &a, int r, int xcol, int xhalo_w, int xhalo_e, int lane, PairLoads &in) {
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int r_s = (r == 0) ? a.ny - 1 : r - 1;
  const int r_n = (r == a.ny - 1) ? 0 : r + 1;
  const float *Rc = a.src + (size_t)r * rs, *Rs = a.src + (size_t)r_s * rs, *Rn = a.src + (size_t)r_n * rs;
  // (wave-uniform row pointer) + (the lane's 32-bit unsigned byte offset): the SGPR-base form of global_load — the plane
  // offsets are scalar additions and no load needs a 64-bit vector address (ten v_lshl_add_u64 per row otherwise); the asm
  // keeps the 32->64-bit extension of the offset next to its uses (instruction selection works per basic block and only
  // then recognises base + zext(offset)), as in issue_row_loads_sbase
  unsigned xb = (unsigned)xcol * 4u;
  asm volatile("" : "+v"(xb));
  auto ld = [&](const float *p) {
    if (NTL) return __builtin_nontemporal_load(reinterpret_cast<const v2f *>(at_byte(p, xb)));
    return *reinterpret_cast<const v2f *>(at_byte(p, xb));
  };
  in.c[0] = ld(Rc); in.c[1] = ld(Rc + ps); in.c[3] = ld(Rc + 3 * ps);
  in.c[2] = ld(Rs + 2 * ps); in.c[5] = ld(Rs + 5 * ps); in.c[6] = ld(Rs + 6 * ps);
  in.c[4] = ld(Rn + 4 * ps); in.c[7] = ld(Rn + 7 * ps); in.c[8] = ld(Rn + 8 * ps);
  // (a whole aligned dword, shifted after loads_ready(): a 16-bit load gets its zero-extension scheduled right behind
  // the load, and the wait for the load with it)
  {
    unsigned mb = (unsigned)(xcol & ~3);
    asm volatile("" : "+v"(mb));
    in.m = *reinterpret_cast<const uint32_t *>(a.mask + (size_t)r * a.nx + mb);
  }
  in.h0 = in.h1 = in.h2 = 0.f;
  if (HALO && (lane == 0 || lane == 63)) {
    const bool lo = (lane == 0);
    in.h0 = lo ? Rc[1 * ps + xhalo_w] : Rc[3 * ps + xhalo_e];
    in.h1 = lo ? Rs[5 * ps + xhalo_w] : Rs[6 * ps + xhalo_e];
    in.h2 = lo ? Rn[8 * ps + xhalo_w] : Rn[7 * ps + xhalo_e];
  }
}

// A window of the deep kernel: slot s holds one v2f per lane at float (s*128 + 2*lane); slots 0..2 = planes 0,1,3 of
// the middle row, slots 3+3p .. 5+3p = the three sweep-direction planes of the rows of parity p (written in iteration
// k, read as the trail row in iteration k+2).  The register form keeps the two trail rows as S0 (older) and S1.
typedef __attribute__((address_space(1))) v2f global_v2f;
constexpr int kPairSlotFloats = 128;
constexpr int kPairWinFloats = 9 * kPairSlotFloats;
struct PairWindow { v2f mid[3], S0[3], S1[3]; };
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
__device__ __forceinline__ v2f lds_pair(const float *p) { return *reinterpret_cast<const v2f *>(p); }
__device__ __forceinline__ v2f lds_pair_shifted(const float *p) {  // 4-byte aligned: one ds_read2_b32
  const f2u t = *reinterpret_cast<const f2u *>(p);
  v2f r = {t.x, t.y};
  return r;
}
__device__ __forceinline__ void lds_pair_put(float *p, v2f v) { *reinterpret_cast<v2f *>(p) = v; }

constexpr int deep_halo_lanes(int D) { return D / 2; }           // ceil((D-1)/2) lanes of two cells at either end of a strip
// Does level 0 need the element beyond the strip's first / last cell (an extra load by lanes 0 and 63)?  Level l's output
// is valid from cell l + (0 with that element, 1 without) inwards; the owned cells start at cell 2*HL: with an even depth
// the halo lanes cover the D levels without it (D = 8: 8 halo cells, levels 0..7 invalidate cells 0..7), an odd depth
// (the twins' D = 5: 4 halo cells) needs it.
constexpr bool deep_edge_loads(int D) { return 2 * deep_halo_lanes(D) < D; }
constexpr int deep_lds_windows(int D) { return D - 1 < 4 ? D - 1 : 4; }  // 4 x 4.5 KB + 1 register window: two waves per SIMD

// The rows level 0 of a sweep works on — rows lo .. hi-1 before wrapping, at most ny of them — hold no blocked cell within
// the strip's lanes (wave-uniform; a handful of scalar loads per chunk).  The deeper levels work on the same rows later.
__device__ __forceinline__ bool strip_rows_clean(const unsigned long long *bits, int words, int strip, int lo, int hi, int ny) {
  if (bits == nullptr || hi - lo >= ny) return false;
  const unsigned long long *w = bits + (size_t)strip * words;
  unsigned long long any = 0;
  auto range = [&](int r0, int r1) {  // 0 <= r0 < r1 <= ny
    for (int i = r0 >> 6; i <= (r1 - 1) >> 6; i++) {
      const int b0 = r0 - 64 * i > 0 ? r0 - 64 * i : 0, b1 = r1 - 64 * i < 64 ? r1 - 64 * i : 64;
      const unsigned long long upto = b1 == 64 ? ~0ull : ((1ull << b1) - 1ull);
      any |= w[i] & upto & ~((1ull << b0) - 1ull);
    }
  };
  int l0 = lo % ny;
  if (l0 < 0) l0 += ny;
  const int l1 = l0 + (hi - lo);
  if (l1 <= ny) range(l0, l1);
  else { range(l0, ny); range(0, l1 - ny); }
  return __builtin_amdgcn_readfirstlane((int)(any == 0ull)) != 0;
}
// Level 0 of a sweep over the chunk [ys, ye) works on these rows (see deep_sweep: r0 + k*d for k = 0 .. last)
__device__ __forceinline__ void deep_sweep_rows(bool up, bool twinned, int L, int ys, int ye, int *lo, int *hi) {
  const int lead = twinned ? 0 : L - 1, count = (ye - ys) + lead + (L - 1);
  const int first = up ? ys - lead : ye - 1 + lead - (count - 1);
  *lo = first;
  *hi = first + count;
}

// The map strip_rows_clean reads: one wave per (strip, word of 64 rows); a lane looks at the mask bytes of its own two
// cells, exactly as issue_pair_loads / deep_sweep address them (lanes beyond the grid's last column wrap).
static __global__ __launch_bounds__(64) void strip_row_bits(const uint8_t *mask, int nx, int rows, int strips, int lanes_out, int halo_lanes,
                                                     unsigned long long *bits, int words) {
  const int strip = blockIdx.x % strips, word = blockIdx.x / strips, lane = threadIdx.x;
  if (word >= words) return;
  const int q2 = nx >> 1;
  int qw = (strip * lanes_out + lane - halo_lanes) % q2;
  if (qw < 0) qw += q2;
  const int xcol = qw * 2;
  unsigned long long out = 0;
  for (int b = 0; b < 64; b++) {
    const int r = word * 64 + b;
    if (r >= rows) break;
    const uint32_t m = *reinterpret_cast<const uint32_t *>(mask + (size_t)r * nx + (xcol & ~3)) >> ((xcol & 2) * 8);
    if (__builtin_amdgcn_ballot_w64((m & 0xffffu) != 0) != 0ull) out |= 1ull << b;
  }
  if (lane == 0) bits[(size_t)strip * words + word] = out;
}

// TWIN (d2q9_deep_twin): the wave is one of the two of a workgroup that work on the chunks 2p (sweeping down) and 2p+1
// (sweeping up) of one strip and START at their common boundary together — the chunk pairs of section 3.5.  A lone wave
// primes its windows with L-1 rows beyond the start of its chunk, which are exactly its neighbour's first rows; twins
// start at their own first row (level l from iteration l on instead of 2l) and write the three planes of their first row
// of every level that move the twin's way into the trail slots of the TWIN's window of that level, where the twin's
// ordinary gather finds them one iteration (and one barrier) later.  n + L-1 iterations per chunk instead of n + 2(L-1).
// Levels whose window lives in registers receive the twin's row through a MAILBOX of three slots behind the wave's
// windows: written in iteration l-1, read in iteration l (early: with the window reads of the level before), and a
// barrier at the start of level l keeps the twin's next write (level l+1, later in the same iteration) behind that read.
// FREE: the caller knows (strip_rows_clean) that none of the rows this sweep works on holds a blocked cell within the
// strip — the whole sweep, start-up included, is the obstacle-free collision: no mask loads, no mask shift chain, no
// ballot and branch per level, and no join of two collision paths for the register allocator to reconcile (the mixed
// steady loop executes 782 VALU instructions per row on its obstacle-free path, this one 709).  Same arithmetic: a wave
// without blocked cells takes collide2<false> either way.
template <int D, int WL, bool UP, bool NT, bool OBST_PATHS, bool TWIN = false, int LT = 0, bool FREE = false>
__device__ __forceinline__ void deep_sweep(const Step2Args &a, const int L_arg, float *lds, float *partials, int pstride, int ys, int ye,
                                           int xcol, int xhalo_w, int xhalo_e, int lane, bool owner, int unit,
                                           const bool twinned_in = false, float *lds_twin = nullptr) {
  // L = timesteps this launch advances (2 .. D, wave-uniform): the run's last launches are shallower.  All row
  // arithmetic is in terms of L; D bounds the unrolled level loop and fixes the halo lanes.  LT > 0: a kernel
  // instantiated for launches of exactly LT timesteps (the host passes nlev == LT).
  static_assert(LT >= 0 && LT <= D, "steady depth");
  const int L = LT > 0 ? LT : L_arg;
  const bool twinned = TWIN && twinned_in;  // wave-uniform; a wave whose twin has no rows runs alone
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };
  const int n = ye - ys, d = UP ? 1 : -1;
  const int lead = twinned ? 0 : L - 1;     // rows before the chunk's first that level 0 starts with
  const int sf = twinned ? 1 : 2;           // level l is active from iteration sf*l on
  const int r0 = UP ? ys - lead : ye - 1 + lead;  // level 0 works on row r0 + k*d in iteration k = 0 .. last,
  const int last = n + lead + (L - 1) - 1;        // level l on row r0 + (k-l)*d
  float sum[D];  // per level: |j|/rho over the lane's two cells (one register per level: a pair each cost 8 VGPRs the steady form needs)
  constexpr int NR = (D - 1 - WL) > 0 ? (D - 1 - WL) : 1;
  PairWindow w[NR];
  uint32_t m_mid[D - 1];
#pragma unroll
  for (int l = 0; l < D; l++) sum[l] = 0.f;
#pragma unroll
  for (int l = 0; l < D - 1; l++) m_mid[l] = 0;
#pragma unroll
  for (int l = 0; l < NR; l++) {
#pragma unroll
    for (int i = 0; i < 3; i++) w[l].mid[i] = w[l].S0[i] = w[l].S1[i] = splat2(0.f);
  }
  float *const lw = lds + 2 + 2 * lane;  // + 2 floats: lane 0 reads one float below its slot
  uint32_t accbits = 0, ownbits = 0;     // bit l: the row level l works on is the accelerated row / one of the chunk's own
  const uint32_t accmask = a.accel_next ? 0xffffffffu : ~(1u << (L - 1));  // the last level's output is the stored state
  const uint32_t mask_shift = (xcol & 2) * 8;  // nx % 4 == 0: the pair's two mask bytes are the low or the high half of a dword
  PairLoads in;
  issue_pair_loads<false, deep_edge_loads(D)>(a, wrap(r0), xcol, xhalo_w, xhalo_e, lane, in);
  // the six window planes level l (1 .. D-1) gathers from: middle row (0, 1 from the west, 3 from the east) and trail row
  float *const mailbox = lw + WL * kPairWinFloats;  // (twins with register windows only)
  auto window_read = [&](int l, int par, int k, v2f (&q)[6]) __attribute__((always_inline)) {
    if ((l - 1) < WL) {
      const float *W = lw + (l - 1) * kPairWinFloats;
      const float *Wp = W + (3 + 3 * par) * kPairSlotFloats;
      q[0] = lds_pair(W); q[1] = lds_pair_shifted(W + kPairSlotFloats - 1); q[2] = lds_pair_shifted(W + 2 * kPairSlotFloats + 1);
      q[3] = lds_pair(Wp); q[4] = lds_pair_shifted(Wp + kPairSlotFloats - 1); q[5] = lds_pair_shifted(Wp + 2 * kPairSlotFloats + 1);
    } else {
      const PairWindow &R = w[(l - 1) < WL ? 0 : (l - 1 - WL)];
      q[0] = R.mid[0]; q[1] = pair_from_west(R.mid[1]); q[2] = pair_from_east(R.mid[2]);
      q[3] = R.S0[0]; q[4] = pair_from_west(R.S0[1]); q[5] = pair_from_east(R.S0[2]);
      if (TWIN && twinned && k == l) {  // the level's first row: its trail row is the twin's first row
        q[3] = lds_pair(mailbox);
        q[4] = lds_pair_shifted(mailbox + kPairSlotFloats - 1);
        q[5] = lds_pair_shifted(mailbox + 2 * kPairSlotFloats + 1);
      }
    }
  };
  v2f out[9];  // the last level's row of the previous iteration
#pragma unroll
  for (int kk = 0; kk < 9; kk++) out[kk] = splat2(0.f);
  auto store_row = [&](int kprev) __attribute__((always_inline)) {
    if (owner) {
      // SGPR-base form again: uniform pointer to the row's plane + the lane's byte offset
      float *const row = a.dst + (size_t)(r0 + (kprev - (L - 1)) * d) * a.row_stride;
      unsigned xb = (unsigned)xcol * 4u;
      asm volatile("" : "+v"(xb));
#pragma unroll
      for (int kk = 0; kk < 9; kk++) {
        // (the plane's row pointer made opaque in scalar registers: otherwise the compiler shares row + offset among the nine
        // stores as ONE 64-bit vector address and adds the plane offsets to it, eight v_lshl_add_u64)
        unsigned long long pk = (unsigned long long)(row + kk * ps);
        asm volatile("" : "+s"(pk));
        global_v2f *const q = (global_v2f *)(pk + xb);  // (a pointer known to be global memory: global_store with an SGPR base)
        if (NT) __builtin_nontemporal_store(out[kk], q);
        else *q = out[kk];
      }
    }
  };
  // One row iteration.  lt_tag 0: the general form — the depth L is a run-time value, a level runs once its first row has
  // arrived (`active`), the level chain ends with a `break`.  lt_tag > 0: the STEADY form for launches of exactly that many
  // timesteps, valid once every level is active (k > sf*(L-1)): depth, activity and "is this the last level" are
  // compile-time facts, so the chain of levels is straight-line code — `nxt` of one level IS `top` of the next (a
  // renaming, where the general form's control flow made the compiler copy nine register pairs per level, ~20 % of its
  // VALU instructions), the last level's row lands in `out` where it is produced, and the start-up branches, the twins'
  // hand-overs and their barriers are gone.  Both forms do the same arithmetic in the same order on every cell.
  auto iteration = [&](const int k, auto lt_tag) __attribute__((always_inline)) {
    constexpr int LS = decltype(lt_tag)::value;
    constexpr bool STEADY = LS > 0;
    const int LL = STEADY ? LS : L;
    const int par = k & 1;
    v2f top[9], pre[2][6];  // pre[l & 1]: the window planes of level l, read one level ahead
    uint32_t m_top;
    const int row0 = wrap(r0 + k * d);
    accbits = ((accbits << 1) | ((row0 == a.accel_row || row0 == a.accel_row_b) ? 1u : 0u)) & accmask;
    ownbits = (ownbits << 1) | ((k >= lead && k <= n + lead - 1) ? 1u : 0u);
    if (STEADY || k >= sf) window_read(1, par, STEADY ? -1 : k, pre[1]);  // issued before level 0's arithmetic: the LDS latency hides behind it
    {  // level 0: step t+1 of row0 from the loaded source rows
      v2f g[9];
      g[0] = in.c[0]; g[2] = in.c[2]; g[4] = in.c[4];
      if (deep_edge_loads(D)) {
        g[1] = pair_from_west(in.c[1], in.h0); g[5] = pair_from_west(in.c[5], in.h1); g[8] = pair_from_west(in.c[8], in.h2);
        g[3] = pair_from_east(in.c[3], in.h0); g[6] = pair_from_east(in.c[6], in.h1); g[7] = pair_from_east(in.c[7], in.h2);
      } else {
        g[1] = pair_from_west(in.c[1]); g[5] = pair_from_west(in.c[5]); g[8] = pair_from_west(in.c[8]);
        g[3] = pair_from_east(in.c[3]); g[6] = pair_from_east(in.c[6]); g[7] = pair_from_east(in.c[7]);
      }
      m_top = FREE ? 0u : in.m >> mask_shift;  // (bits 16.. may hold the neighbouring pair's bytes: every test masks)
      v2f t;
      if (FREE || (OBST_PATHS && __builtin_amdgcn_ballot_w64((m_top & 0xffffu) != 0) == 0ull)) t = collide2<false>(g, m_top, a.omega, (accbits & 1u) != 0, a.aw1, a.aw2, top);
      else t = collide2<true>(g, m_top, a.omega, (accbits & 1u) != 0, a.aw1, a.aw2, top);
      if ((ownbits & 1u) && owner) sum[0] += t.x + t.y;
      // The row the last level finished in the PREVIOUS iteration is stored here, right before this iteration's loads:
      // the wave waits for its loads at the top of the next iteration with the memory counter at zero, stores
      // included — with the stores at the end of an iteration that wait exposed the round trip of stores just issued;
      // now everything it covers was issued a whole iteration of arithmetic earlier.
      // (stores and loads issued BEFORE level 0's arithmetic, right after its gather: 1 % slower everywhere, tools/ab.py --libs)
      if (STEADY || k - 1 >= sf * (LL - 1)) store_row(k - 1);
      // (unconditional: the last iteration loads its own row once more rather than branching around the loads)
      issue_pair_loads<false, deep_edge_loads(D)>(a, wrap(r0 + (k < last ? k + 1 : k) * d), xcol, xhalo_w, xhalo_e, lane, in);
    }
#pragma unroll
    for (int l = 1; l < D; l++) {
      if (STEADY && l >= LS) break;  // (compile-time: the unrolled chain ends here)
      if (STEADY) __builtin_amdgcn_sched_barrier(0);  // one level after the other: interleaved levels cost hundreds of spilled registers
      const bool final = (l == D - 1) || (l == LL - 1);  // (level l exists: the level before it was not the last)
      if (!STEADY && TWIN && (l - 1) >= WL && twinned && k == l) __syncthreads();  // mailbox read (above) before the twin's next write
      v2f nxt[9];
      uint32_t m_nxt = 0;
      const bool active = STEADY || k >= sf * l;
      const bool in_lds = (l - 1) < WL;
      float *const W = lw + (l - 1) * kPairWinFloats;
      float *const Wp = W + (3 + 3 * par) * kPairSlotFloats;
      PairWindow &R = w[in_lds ? 0 : (l - 1 - WL)];
      if (active) {
        const v2f (&q)[6] = pre[l & 1];
        if (l + 1 < D && !final && (STEADY || k >= sf * (l + 1))) window_read(l + 1, par, STEADY ? -1 : k, pre[(l + 1) & 1]);  // the next level's window, early
        v2f g[9];
        g[0] = q[0]; g[1] = q[1]; g[3] = q[2];
        if (UP) {  // the trail row is the row below: its planes 2,5,6 arrive; the newest row is above: 4,7,8
          g[2] = q[3]; g[5] = q[4]; g[6] = q[5];
          g[4] = top[4]; g[8] = pair_from_west(top[8]); g[7] = pair_from_east(top[7]);
        } else {
          g[4] = q[3]; g[8] = q[4]; g[7] = q[5];
          g[2] = top[2]; g[5] = pair_from_west(top[5]); g[6] = pair_from_east(top[6]);
        }
        m_nxt = FREE ? 0u : m_mid[l - 1];
        const bool acc = ((accbits >> l) & 1u) != 0;
        v2f t;
        if (FREE || (OBST_PATHS && __builtin_amdgcn_ballot_w64((m_nxt & 0xffffu) != 0) == 0ull)) t = collide2<false>(g, m_nxt, a.omega, acc, a.aw1, a.aw2, nxt);
        else t = collide2<true>(g, m_nxt, a.omega, acc, a.aw1, a.aw2, nxt);
        if (!final) {
          if (((ownbits >> l) & 1u) && owner) sum[l] += t.x + t.y;
        } else {  // the last level: every row it works on is one of the chunk's own; stored in the NEXT iteration
          if (owner) sum[l] += t.x + t.y;
#pragma unroll
          for (int kk = 0; kk < 9; kk++) out[kk] = nxt[kk];
        }
      }
      // `top` becomes the window's middle row, its sweep-direction planes the newest trail row
      if (in_lds) {
        lds_pair_put(W, top[0]); lds_pair_put(W + kPairSlotFloats, top[1]); lds_pair_put(W + 2 * kPairSlotFloats, top[3]);
        lds_pair_put(Wp, UP ? top[2] : top[4]);
        lds_pair_put(Wp + kPairSlotFloats, UP ? top[5] : top[8]);
        lds_pair_put(Wp + 2 * kPairSlotFloats, UP ? top[6] : top[7]);
      } else {
        R.mid[0] = top[0]; R.mid[1] = top[1]; R.mid[2] = top[3];
#pragma unroll
        for (int i = 0; i < 3; i++) R.S0[i] = R.S1[i];
        R.S1[0] = UP ? top[2] : top[4]; R.S1[1] = UP ? top[5] : top[8]; R.S1[2] = UP ? top[6] : top[7];
      }
      if (!FREE) m_mid[l - 1] = m_top;
      if (!STEADY && TWIN && twinned && k == l - 1) {
        // `top` is the first row of level l-1: its planes that move the twin's way become the trail row of the twin's
        // first gather of level l, next iteration (the twin reads parity l & 1 then; its own puts reach that slot later;
        // a register window: through the twin's mailbox)
        float *const Wt = lds_twin + 2 + 2 * lane + (in_lds ? (l - 1) * kPairWinFloats + (3 + 3 * (l & 1)) * kPairSlotFloats : WL * kPairWinFloats);
        lds_pair_put(Wt, UP ? top[4] : top[2]);
        lds_pair_put(Wt + kPairSlotFloats, UP ? top[8] : top[5]);
        lds_pair_put(Wt + 2 * kPairSlotFloats, UP ? top[7] : top[6]);
      }
      if (!active || final) break;
      if (l < D - 1) {
#pragma unroll
        for (int kk = 0; kk < 9; kk++) top[kk] = nxt[kk];
        m_top = m_nxt;
      }
    }
    if (!STEADY && TWIN && twinned && k <= L - 2) __syncthreads();  // both twins run these iterations; the hand-over of iteration k is read in k+1
  };
  // Start-up in the general form (level l joins in iteration sf*l), then the steady form to the end of the chunk.  The
  // steady form exists in kernels instantiated for ONE depth (LT > 0: the launch advances exactly LT timesteps, L is
  // that constant); LT = 0 is the kernel for any depth, general form throughout.  (Tried and dropped: all depths in one
  // kernel — one loop per form: what the later loops hoist is live through the earlier ones, 600 spilled registers;
  // one loop that picks its body per iteration: 44-206.)
  const int k_steady = sf * (L - 1) + 1;
  int k = 0;
  if (LT > 0) {
    for (; k < k_steady && k <= last; k++) iteration(k, std::integral_constant<int, 0>{});
    for (; k <= last; k++) iteration(k, std::integral_constant<int, LT>{});
  } else {
    for (; k <= last; k++) iteration(k, std::integral_constant<int, 0>{});
  }
  store_row(last);
#pragma unroll
  for (int l = 0; l < D; l++) {
    if (l >= L) break;
    const float s = wave_sum(sum[l]);
    if (lane == 0) partials[(size_t)l * pstride + unit] = s;
  }
}

// nlev = timesteps the launch advances (2 .. D); partials: [nlev][pstride], slot l*pstride + unit = the unit's sum of
// |j|/rho after step t+1+l
// An edge unit's output rows a second time, into the ring neighbour's halo rows (push_chunk for lanes of two cells: one
// 8-byte write-through store per plane and lane).
__device__ __forceinline__ void push_chunk_pairs(const Step2Args *la, int ys, int ye, int xcol, bool owner) {
  drain_stores();
  const HaloPeer *pp = uniform_ptr(la->peer);
  const int hi0 = __builtin_amdgcn_readfirstlane(pp->row_hi0);
  const int lo0 = __builtin_amdgcn_readfirstlane(pp->row_lo0);
  const int rows = __builtin_amdgcn_readfirstlane(pp->push_rows);
  const int buf = __builtin_amdgcn_readfirstlane(la->peer_buf) & 1;
  const int side = ys >= hi0 ? 1 : 0;
  const int base = side ? hi0 : lo0;
  if (ys < base || ye > base + rows) {
    if ((threadIdx.x & 63) == 0) atomicOr(pp->wait_err, 2u);
    return;
  }
  const size_t rs = la->row_stride, ps = la->plane_stride;
  const float *own = la->dst + xcol;
  float *peer = pp->push[side][buf] + xcol;
  if (owner) {
    for (int y = ys; y < ye; y++) {
      const float *src = own + (size_t)y * rs;
      float *dst = peer + (size_t)(y - base) * rs;
      v2f v[9];
#pragma unroll
      for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const v2f *>(src + k * ps);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        global_u64 *q = (global_u64 *)(unsigned long long)(dst + k * ps);
        const unsigned long long bits = ((unsigned long long)__float_as_uint(v[k].y) << 32) | __float_as_uint(v[k].x);
        __hip_atomic_store(q, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  release_pushed(__builtin_amdgcn_readfirstlane(pp->release));
}

// PUSH: the kernel as ONE launch per launch set of a slab (compact launch sets, see Step2Args): the first edge_units
// workgroups work through the edge schedule, push their rows into the ring neighbours and raise the flag words.
template <int D, bool NT, bool OBST_PATHS = false, bool PUSH = false, int LT = 0>
__global__ __launch_bounds__(64, 2) void d2q9_deep(const Step2Args a, float *partials, int pstride, int nlev) {
  constexpr int WL = deep_lds_windows(D), HL = deep_halo_lanes(D);
  __shared__ float lds[WL * kPairWinFloats + 4];
  const int lane = threadIdx.x;
  const UnitSel us = select_unit<PUSH>(a, a.edge_units);
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if (lane < 2) spin_on_flag(pp->wait_flags + lane, la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;
  const int unit0 = band * us.units_per_band + slot;
  const int sdiv = (PUSH && us.edge) ? a.strips_edge : a.strips;
  const int chunk = unit0 / sdiv;
  int strip = unit0 - chunk * sdiv;
  const int unit = unit0 + us.partial_off;  // slot of the velocity sums
  int ys, ye;
  if (a.vmap != nullptr && !(PUSH && us.edge)) {  // (virtual strips, see Step2Args)
    const int *e = a.vtab + 2 * ((size_t)a.vmap[2 * strip + 1] * a.nchunks + chunk);
    strip = a.vmap[2 * strip];
    ys = e[0];
    ye = e[1];
  } else {
    ys = us.chunk_start[chunk];
    ye = us.chunk_start[chunk + 1];
  }
  if (ys >= ye || chunk == us.skip) {
    if (lane < nlev) partials[(size_t)lane * pstride + unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;
  }
  const int q2 = a.nx >> 1;
  const int qcol = strip * a.lanes_out + lane - HL;
  const bool owner = (lane >= HL) && (lane < HL + a.lanes_out) && (qcol < q2);
  int qw = qcol % q2;
  if (qw < 0) qw += q2;
  const int xcol = qw * 2;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 2 >= a.nx) ? 0 : xcol + 2;
  // even chunks sweep up, odd chunks down: neighbouring chunks meet at their common boundary rows at about the same time
  const bool up = __builtin_amdgcn_readfirstlane((int)((chunk & 1) == 0)) != 0;
  bool clean = false;
  if constexpr (LT > 0 && OBST_PATHS) {  // (the kernels of the default configuration)
    int lo, hi;
    deep_sweep_rows(up, false, LT, ys, ye, &lo, &hi);
    clean = strip_rows_clean(a.clean_bits, a.clean_words, strip, lo, hi, a.ny);
  }
  if (clean) {
    if constexpr (LT > 0 && OBST_PATHS) {
      if (up) deep_sweep<D, WL, true, NT, OBST_PATHS, false, LT, true>(a, nlev, lds, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit);
      else deep_sweep<D, WL, false, NT, OBST_PATHS, false, LT, true>(a, nlev, lds, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit);
    }
  } else if (up) {
    deep_sweep<D, WL, true, NT, OBST_PATHS, false, LT>(a, nlev, lds, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit);
  } else {
    deep_sweep<D, WL, false, NT, OBST_PATHS, false, LT>(a, nlev, lds, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit);
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      push_chunk_pairs(la, ys, ye, xcol, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// Chunk pairs of d2q9_deep: a workgroup is two waves, the chunks 2p (down) and 2p+1 (up) of one strip (see deep_sweep).
// units_per_band counts chunk PAIRS x strips.  LDS per wave: the four windows + the mailbox = 19.98 KB: four workgroups
// (eight waves) still fit a CU.
// PUSH: the compact launch-set form for a slab that exchanges halo rows (as d2q9_deep<..., PUSH>): the first edge_units / 2
// workgroups are the EDGE workgroups — wave 0 works on the bottom edge rows, wave 1 on the top edge rows of one strip (chunks 0
// and 2 of the edge table {bottom edge, (interior), top edge}); neither has a twin, both run alone, push their rows into
// the ring neighbours, and the last edge wave raises the flag words —, the others are the interior's chunk pairs.
// PUSH with D <= 5 (slabs of 240K to 3M cells, five halo rows): the edge rows of such a slab are as long a sweep as an interior
// chunk is (2-5 rows), so a lone edge wave — H + 2(D-1) = 13 iterations against the pairs' ~7 — would be what the launch waits
// for.  Here the H rows at either end are a chunk PAIR of their own (edge table {b0, b1, b2 | interior | t0 (empty), t0, t1, t2}:
// chunks 0/1 and 4/5; the first 2 x strips workgroups), twinned like any other; every edge wave pushes its own rows.
template <int D, bool NT, bool OBST_PATHS = false, int LT = 0, bool PUSH = false>
__global__ __launch_bounds__(128, 2) void d2q9_deep_twin(const Step2Args a, float *partials, int pstride, int nlev) {
  constexpr int WL = deep_lds_windows(D), HL = deep_halo_lanes(D);
  constexpr int kWaveFloats = WL * kPairWinFloats + (D - 1 > WL ? 3 * kPairSlotFloats : 0) + 4;
  constexpr bool EDGE_PAIRS = PUSH && D <= 5;
  __shared__ float lds[2 * kWaveFloats];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const UnitSel us = select_unit<PUSH>(a, a.edge_units / 2);  // edge_units counts edge WAVES
  const bool do_push = PUSH && us.edge && (a.peer_mode & 1);
  if constexpr (PUSH) {
    // in-kernel wait for the neighbours' halo rows (halo_sync = 2): every edge wave for itself, before its first load
    if (us.edge && (a.peer_mode & 2)) {
      const Step2Args *la = late_args<Step2Args>();
      const HaloPeer *pp = la->peer;
      if (lane < 2) spin_on_flag(pp->wait_flags + lane, la->wait_seq, pp->wait_err, pp->wait_ticks);
    }
  }
  const int band = us.bid % us.nbands, slot = us.bid / us.nbands;
  if (slot >= us.units_per_band) return;  // units_per_band counts workgroups: chunk PAIRS x strips (edge: strips)
  const int punit = band * us.units_per_band + slot;
  const bool edge_wg = PUSH && us.edge;
  const int sdiv = edge_wg ? a.strips_edge : a.strips;
  const int pair = punit / sdiv;
  int strip = punit - pair * sdiv;
  const int chunk = edge_wg ? (EDGE_PAIRS ? 4 * pair + wv : 2 * wv) : 2 * pair + wv;
  const int unit = ((edge_wg && EDGE_PAIRS) ? 2 * pair + wv : chunk) * sdiv + strip + us.partial_off;  // (edge pairs: four slots per strip)
  int ys, ye, pys, pye;
  if (a.vmap != nullptr && !edge_wg) {  // (virtual strips, see Step2Args: a copy's chunks 2p / 2p+1 are the halves of one chunk)
    const int *e = a.vtab + 2 * ((size_t)a.vmap[2 * strip + 1] * a.nchunks + chunk), *pe = e + (wv ? -2 : 2);
    strip = a.vmap[2 * strip];
    ys = e[0]; ye = e[1];
    pys = pe[0]; pye = pe[1];
  } else {
    ys = us.chunk_start[chunk]; ye = us.chunk_start[chunk + 1];
    pys = us.chunk_start[chunk ^ 1]; pye = us.chunk_start[(chunk ^ 1) + 1];
  }
  const bool empty = ys >= ye || (!(PUSH && us.edge) && chunk == us.skip);
  const bool twinned = !(PUSH && us.edge && !EDGE_PAIRS) && !empty && pys < pye && (chunk ^ 1) != us.skip;  // the same on both waves
  if constexpr (PUSH && !EDGE_PAIRS) {
    // nobody works on the edge table's middle chunk (the interior): its slot of the velocity sums is this wave's to clear
    if (us.edge && wv == 0 && lane < nlev) partials[(size_t)lane * pstride + (a.strips_edge + strip + us.partial_off)] = 0.f;
  }
  if (empty) {
    if (lane < nlev) partials[(size_t)lane * pstride + unit] = 0.f;
    if constexpr (PUSH) {
      const Step2Args *la = late_args<Step2Args>();
      if (us.edge && (la->peer_mode & 1)) publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
    return;  // the twin then runs alone and meets no barrier
  }
  const int q2 = a.nx >> 1;
  const int qcol = strip * a.lanes_out + lane - HL;
  const bool owner = (lane >= HL) && (lane < HL + a.lanes_out) && (qcol < q2);
  int qw = qcol % q2;
  if (qw < 0) qw += q2;
  const int xcol = qw * 2;
  const int xhalo_w = (xcol == 0) ? a.nx - 1 : xcol - 1;
  const int xhalo_e = (xcol + 2 >= a.nx) ? 0 : xcol + 2;
  float *const mine = lds + wv * kWaveFloats, *const theirs = lds + (wv ^ 1) * kWaveFloats;
  // odd chunks sweep up from their bottom row, even ones down from their top row: twins start together.  Each wave picks
  // its own form of the sweep (both forms meet the same barriers)
  bool clean = false;
  if constexpr (LT > 0 && OBST_PATHS && D == 8) {
    int lo, hi;
    deep_sweep_rows(wv != 0, twinned, LT, ys, ye, &lo, &hi);
    clean = strip_rows_clean(a.clean_bits, a.clean_words, strip, lo, hi, a.ny);
  }
  if (clean) {
    if constexpr (LT > 0 && OBST_PATHS && D == 8) {
      if (wv != 0) deep_sweep<D, WL, true, NT, OBST_PATHS, true, LT, true>(a, nlev, mine, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit, twinned, theirs);
      else deep_sweep<D, WL, false, NT, OBST_PATHS, true, LT, true>(a, nlev, mine, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit, twinned, theirs);
    }
  } else if (wv != 0) {
    deep_sweep<D, WL, true, NT, OBST_PATHS, true, LT>(a, nlev, mine, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit, twinned, theirs);
  } else {
    deep_sweep<D, WL, false, NT, OBST_PATHS, true, LT>(a, nlev, mine, partials, pstride, ys, ye, xcol, xhalo_w, xhalo_e, lane, owner, unit, twinned, theirs);
  }
  if constexpr (PUSH) {
    const Step2Args *la = late_args<Step2Args>();
    if (do_push) {
      push_chunk_pairs(la, ys, ye, xcol, owner);
      publish_wave_when_last(la->peer, (unsigned)la->edge_units, la->seq);
    }
  }
}

// ---- all timesteps of a run in ONE launch, the grid resident in registers ("resident" kernel) ------------------
// The latency-bound sizes (512x512 ... 1024x1024: the sizes the reference publishes) pay, in every kernel above, a pass
// through memory per launch and a launch per <= 8 steps: d2q9_deep_twin<5> runs 1024x1024 at 5.8 us/step, of which ~2.4 us is
// arithmetic.  Here the grid never leaves the chip between the steps of an lbm_run: 1024x1024 cells x 36 B = 37.7 MB against
// 128 MB of vector registers.  A WORKGROUP of W waves owns a band of BH (2, 4 or 6) full-width rows (W = nx / 128: a wave holds 64 lanes of
// two cells x BH rows x 9 planes = 18 BH registers), one workgroup per CU at W = 8, all of them co-resident; per timestep
//   - the waves of a band trade their edge cells (x neighbours, periodic wrap included) through LDS: one barrier;
//   - a wave stores the three planes of its top row that move up and of its bottom row that move down into the band's
//     exchange rows in memory (double-buffered by step parity; 8-byte agent-scope write-through stores, drained), then
//     raises its step word; the rows between (BH - 2 of them) need nothing from outside and are collided while the words
//     travel; then the wave polls the step words of the three waves above and the three below it (bounded spin, one timeout
//     per run), loads their rows and collides its bottom and top row.
// No launch boundary, no start-up rows, no halo lanes, no redundant cell: 2 cells x 88 packed instructions per lane, row and
// step — and one neighbour hand-shake per step on the critical path (tools/resident_probe.cpp: 2.2 us, 2.85 us per step with the
// arithmetic of four rows; two steps per hand-shake on two-row halos come out the same).  In-place update: the rows are
// collided in the order 1 .. BH-2, 0, BH-1 and the few planes a later row still needs of an overwritten one are kept aside.
// Same collide2 as d2q9_deep on the same pairs of cells: bit-identical to single steps.
struct ResidentArgs {
  const float *src;          // the grid at the first step (row-interleaved planes, as everywhere)
  float *dst;                // ... after the last
  const uint8_t *mask;
  float *partials;           // [nsteps][pstride]: slot band * W + wave = that wave's sum of |j|/rho per step
  unsigned long long plane_stride, row_stride, pstride;
  int nx, ny, nsteps;
  int accel_row;             // global row ny-2 (-1: none)
  int accel_next;            // apply the following step's accelerate_flow after the LAST step too
  float omega, aw1, aw2;
  unsigned *words;           // [bands][32]: a band's W step words in one 128-byte line
  float *xrows;              // [2 parity][bands][2 dir][3 planes][nx]: dir 0 = a band's top row (planes 2,5,6), dir 1 = its bottom row (4,7,8)
  unsigned seq_base;         // the words reach seq_base + 1 + (steps done in this launch)
  unsigned *err;
  unsigned long long wait_ticks;
};

__device__ __forceinline__ void resident_store(float *p, v2f v) {
  global_u64 *q = (global_u64 *)(unsigned long long)p;
  const unsigned long long bits = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
  __hip_atomic_store(q, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ v2f resident_load(const float *p) {
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v2f v = {__uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
  return v;
}
__device__ __forceinline__ float resident_load1(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// OBST: the wave holds a blocked cell (the obstacle map does not change during a run).  The whole band loop exists in both forms and
// the kernel chooses ABOVE it: chosen per row inside the loop, the compiler merged the two collision paths into one that carries
// the bounce-back selects and copies every row's result into place (916 VALU instructions per step of four rows, 251 of them moves,
// 154 selects; the obstacle-free loop below has 555).  Both forms meet the same barriers.
// PARTIAL: the band's last wave where nx is no multiple of 128 — its lanes from `nl` on hold no cell (they work on a copy of the last
// pair and store nothing), and the "lane 63" of the x exchange is lane nl-1.
template <int BH, bool OBST, bool PARTIAL>
__device__ __forceinline__ void resident_band(const ResidentArgs &a, float (&xe)[2][8][BH][4], float (&xw)[2][8][BH][4]) {
  const int W = (int)(blockDim.x >> 6);   // waves across the band = ceil(nx / 128) (1 .. 8): a launch parameter, not a template one
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int band = blockIdx.x, nbands = gridDim.x;
  const int x0 = w * 128;
  const int nl = PARTIAL ? (a.nx - x0) / 2 : 64;           // lanes of this wave that hold cells
  const bool active = !PARTIAL || lane < nl;
  const int xcol = PARTIAL ? (lane < nl ? x0 + 2 * lane : a.nx - 2) : x0 + 2 * lane;   // (idle lanes read the row's last pair)
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int wl = (w + W - 1) % W, wr = (w + 1) % W;
  const int bdn = (band + nbands - 1) % nbands, bup = (band + 1) % nbands;
  v2f f[BH][9];
  uint32_t m[BH];
#pragma unroll
  for (int r = 0; r < BH; r++) {
    const size_t row = (size_t)(band * BH + r);
#pragma unroll
    for (int k = 0; k < 9; k++) f[r][k] = *reinterpret_cast<const v2f *>(a.src + row * rs + k * ps + xcol);
    m[r] = OBST ? (*reinterpret_cast<const uint32_t *>(a.mask + row * a.nx + (xcol & ~3)) >> ((xcol & 2) * 8)) & 0xffffu : 0u;
  }
  const size_t xr_dir = (size_t)3 * a.nx, xr_band = 2 * xr_dir, xr_par = (size_t)nbands * xr_band;
  // (a band's W step words share one 128-byte line: the three words a wave polls per side are one memory transaction)
  unsigned *const my_word = a.words + (size_t)band * 32 + w;
  // the step words this wave waits for: lanes 0..2 the band below (waves wl, w, wr), lanes 3..5 the band above
  const unsigned *poll = a.words + (size_t)(lane < 3 ? bdn : bup) * 32 + (lane % 3 == 0 ? wl : (lane % 3 == 1 ? w : wr));
  auto publish = [&](unsigned seq) __attribute__((always_inline)) {
    float *base = a.xrows + (size_t)(seq & 1u) * xr_par + (size_t)band * xr_band + xcol;
    if (!active) return;
    resident_store(base + 0 * a.nx, f[BH - 1][2]);
    resident_store(base + 1 * a.nx, f[BH - 1][5]);
    resident_store(base + 2 * a.nx, f[BH - 1][6]);
    resident_store(base + xr_dir + 0 * a.nx, f[0][4]);
    resident_store(base + xr_dir + 1 * a.nx, f[0][7]);
    resident_store(base + xr_dir + 2 * a.nx, f[0][8]);
  };
  auto raise_word = [&](unsigned seq) __attribute__((always_inline)) {
    drain_stores();  // write-through stores: acknowledged = they have left this XCD
    if (lane == 0) __hip_atomic_store(my_word, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  publish(a.seq_base + 1u);
  raise_word(a.seq_base + 1u);
  for (int s = 0; s < a.nsteps; s++) {
    const unsigned seq = a.seq_base + 1u + (unsigned)s;   // the neighbours' rows of the state this step starts from
    const int par = s & 1;
    const bool accel_ok = (s + 1 < a.nsteps) || a.accel_next;
    // ---- x neighbours through LDS
    if (lane == nl - 1) {
#pragma unroll
      for (int r = 0; r < BH; r++) { xe[par][w][r][0] = f[r][1].y; xe[par][w][r][1] = f[r][5].y; xe[par][w][r][2] = f[r][8].y; }
    }
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < BH; r++) { xw[par][w][r][0] = f[r][3].x; xw[par][w][r][1] = f[r][6].x; xw[par][w][r][2] = f[r][7].x; }
    }
    __syncthreads();
    float hw[BH][3], he[BH][3];  // west halos of planes 1,5,8 / east halos of planes 3,6,7, by the row they are stored in
#pragma unroll
    for (int r = 0; r < BH; r++) {
#pragma unroll
      for (int k = 0; k < 3; k++) { hw[r][k] = xe[par][wl][r][k]; he[r][k] = xw[par][wr][r][k]; }
    }
    float sum = 0.f;
    // (the pair shifted in from the east: lane 63 — in a partial wave lane nl-1 — takes the neighbour wave's first cell)
    auto from_east = [&](v2f p, float halo) __attribute__((always_inline)) {
      v2f r = pair_from_east(p, halo);
      if (PARTIAL) r.y = (lane == nl - 1) ? halo : r.y;
      return r;
    };
    // one row: south = planes 2,5,6 of the row below (+ their x halos: west of 5, east of 6), north = planes 4,7,8 of the row above
    // (west of 8, east of 7)
    auto collide_row = [&](int r, const v2f (&south)[3], float s5w, float s6e, const v2f (&north)[3], float n8w, float n7e) __attribute__((always_inline)) {
      v2f g[9], o[9];
      g[0] = f[r][0];
      g[1] = pair_from_west(f[r][1], hw[r][0]);
      g[3] = from_east(f[r][3], he[r][0]);
      g[2] = south[0];
      g[5] = pair_from_west(south[1], s5w);
      g[6] = from_east(south[2], s6e);
      g[4] = north[0];
      g[7] = from_east(north[1], n7e);
      g[8] = pair_from_west(north[2], n8w);
      const bool acc = accel_ok && (band * BH + r == a.accel_row);
      v2f t;
      if constexpr (!OBST) t = collide2<false>(g, 0u, a.omega, acc, a.aw1, a.aw2, o);
      else t = collide2<true>(g, m[r], a.omega, acc, a.aw1, a.aw2, o);
      sum += active ? t.x + t.y : 0.f;
#pragma unroll
      for (int k = 0; k < 9; k++) f[r][k] = o[k];
    };
    // the planes later rows still need of rows that get overwritten first
    v2f prev256[3] = {f[0][2], f[0][5], f[0][6]};      // old row r-1, for row r
    float prev5w = hw[0][1], prev6e = he[0][1];
    v2f one478[3] = {f[BH > 2 ? 1 : 0][4], f[BH > 2 ? 1 : 0][7], f[BH > 2 ? 1 : 0][8]};  // old row 1, for row 0 (BH = 2: unused)
#pragma unroll
    for (int r = 1; r <= BH - 2; r++) {
      const v2f keep[3] = {f[r][2], f[r][5], f[r][6]};
      const v2f north[3] = {f[r + 1][4], f[r + 1][7], f[r + 1][8]};
      collide_row(r, prev256, prev5w, prev6e, north, hw[r + 1][2], he[r + 1][2]);
#pragma unroll
      for (int k = 0; k < 3; k++) prev256[k] = keep[k];
      prev5w = hw[r][1];
      prev6e = he[r][1];
    }
    // (prev256 now holds the old row BH-2 — for BH = 2 the old row 0 — which row BH-1 reads)
    // ---- y neighbours: wait for the step words, load the rows
    if (lane < 6) {
      // bounded spin; ONE timeout per run (a wait that has lasted 100 us looks at the error word other waits may have raised — not
      // before: six lanes of 2048 waves reading one word on every step would make that word's memory channel the hand-shake)
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((int32_t)(__hip_atomic_load(poll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
        __builtin_amdgcn_s_sleep(1);
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
        if (dt > a.wait_ticks) { atomicOr(a.err, 1u); break; }
        if (dt > 10000ull && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const float *below = a.xrows + (size_t)(seq & 1u) * xr_par + (size_t)bdn * xr_band;           // its top row: planes 2,5,6
    const float *above = a.xrows + (size_t)(seq & 1u) * xr_par + (size_t)bup * xr_band + xr_dir;  // its bottom row: planes 4,7,8
    v2f hb[3], ha[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { hb[k] = resident_load(below + k * a.nx + xcol); ha[k] = resident_load(above + k * a.nx + xcol); }
    // the corner elements: one cell beyond the wave's 128 (only lanes 0 and 63 use them)
    const int xwest = (x0 + a.nx - 1) % a.nx, xeast = (x0 + 2 * nl) % a.nx;
    float b5w = 0.f, b6e = 0.f, a8w = 0.f, a7e = 0.f;
    if (lane == 0) { b5w = resident_load1(below + 1 * a.nx + xwest); a8w = resident_load1(above + 2 * a.nx + xwest); }
    if (lane == nl - 1) { b6e = resident_load1(below + 2 * a.nx + xeast); a7e = resident_load1(above + 1 * a.nx + xeast); }
    if constexpr (BH > 2) {
      collide_row(0, hb, b5w, b6e, one478, hw[1][2], he[1][2]);
    } else {
      const v2f north[3] = {f[1][4], f[1][7], f[1][8]};
      collide_row(0, hb, b5w, b6e, north, hw[1][2], he[1][2]);
    }
    collide_row(BH - 1, prev256, prev5w, prev6e, ha, a8w, a7e);
    if (s + 1 < a.nsteps) publish(seq + 1u);
    const float tot = wave_sum(sum);  // (between the stores and their drain: the reduction runs while the stores are acknowledged)
    if (s + 1 < a.nsteps) raise_word(seq + 1u);
    if (lane == 0) a.partials[(size_t)s * a.pstride + band * W + w] = tot;
  }
  if (active) {
#pragma unroll
    for (int r = 0; r < BH; r++) {
      const size_t row = (size_t)(band * BH + r);
#pragma unroll
      for (int k = 0; k < 9; k++) *reinterpret_cast<v2f *>(a.dst + row * rs + k * ps + xcol) = f[r][k];
    }
  }
}

template <int BH>
__global__ __launch_bounds__(512) void d2q9_resident(const ResidentArgs a) {
  static_assert(BH >= 2 && BH <= 6, "band shape");
  // x exchange inside the band: [parity][wave][row][3]: east[.] = planes 1,5,8 of the wave's last cell, west[.] = 3,6,7 of its first
  __shared__ float xe[2][8][BH][4], xw[2][8][BH][4];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool partial = w * 128 + 128 > a.nx;   // the band's last wave where nx is no multiple of 128
  const int xcol = w * 128 + 2 * lane < a.nx ? w * 128 + 2 * lane : a.nx - 2;
  uint32_t any = 0;
#pragma unroll
  for (int r = 0; r < BH; r++)
    any |= (*reinterpret_cast<const uint32_t *>(a.mask + (size_t)(blockIdx.x * BH + r) * a.nx + (xcol & ~3)) >> ((xcol & 2) * 8)) & 0xffffu;
  const bool obst = __builtin_amdgcn_ballot_w64(any != 0u) != 0ull;
  if (partial) {
    if (obst) resident_band<BH, true, true>(a, xe, xw);
    else resident_band<BH, false, true>(a, xe, xw);
  } else if (obst) {
    resident_band<BH, true, false>(a, xe, xw);
  } else {
    resident_band<BH, false, false>(a, xe, xw);
  }
}

// ---- T timesteps per launch on an LDS-resident tile (small grids) ---------------------------------
// Grids of a few hundred cells a side are bound by launch latency, not bandwidth (one step of 128x128 is
// ~2 us of work behind ~3.4 us of launch cost).  This kernel advances T <= kMultiMaxT steps per launch: a
// workgroup loads a (TX + 2T) x (TY + 2T) region around its TX x TY output tile into LDS, performs T
// stream+collide steps LDS -> LDS on a region that shrinks by one cell per step (halo cells are computed
// redundantly by the neighbouring tiles), and stores the central tile.  Same per-cell arithmetic
// (collide_cell / accelerate_cell) as the other kernels, so the results are bit-identical.
constexpr int kMultiMaxT = 8;
constexpr int kMultiThreads = 1024;
// Output tile TX x TY (template): 32x16 when the grid has enough tiles to fill the CUs, 16x16 or 16x8 for the
// smallest grids — a sub-step is VALU-bound on the tile's one CU (~110 instructions per cell on 4 SIMDs), so a
// grid that cannot fill 256 CUs anyway runs faster on more, smaller tiles despite their larger share of halo cells.

struct MultiArgs {
  const float *src;
  float *dst;
  const uint8_t *mask;
  float *partials;          // [T][partials_stride]: per-tile sums of |j|/rho for each of the T steps
  unsigned long long plane_stride, row_stride, partials_stride;
  int nx;
  int rows;                 // owned rows (the whole grid with one slab)
  int ext_rows;             // rows stored (owned + halo rows below and above)
  int row_off;              // stored row of owned row 0; 0 = no halo rows: y wraps periodically inside the kernel
  int tiles_x;
  int ty_begin, ty_split, ty_begin2;  // tile row of workgroup-row t: t < ty_split ? ty_begin + t : ty_begin2 + (t - ty_split) ...
  int ty_split2, ty_begin3;           // ... and, PEER form only, ty_begin3 + (t - ty_split2) from t = ty_split2 on
  int T;                    // steps in this launch (<= row_off when there are halo rows)
  int gy_off, ny_global;    // stored row r holds global row (gy_off + r) mod ny_global; accelerate_flow acts on ny_global-2
  int accel_next;           // apply the following step's accelerate_flow to the final state
  float omega, aw1, aw2;
  // peer-halo transport, fused ("compact" launch sets of small slabs): the first edge_blocks workgroups hold the
  // bottom and top owned rows the ring neighbours need; they store those rows a second time, write-through, into the
  // neighbours' halo rows, and the last of them to finish raises the neighbours' flag words to seq (peer_mode bit 0).
  // Bit 1 (option halo_sync = 2): the same workgroups — the only ones whose regions reach into the halo rows — poll
  // this slab's own flag words for wait_seq before they load (bounded, like halo_wait).  Everything that does not
  // change from launch to launch sits in the device-resident HaloPeer (384x384 ran 4.54 instead of 3.46 us/step with
  // those pointers as kernel arguments).
  const struct HaloPeer *peer;
  uint32_t seq, wait_seq;
  int peer_mode, peer_buf, edge_blocks;   // peer_buf: index of the destination grid (which of the neighbours' two grids)
};

// PEER: the compact launch-set form (three tile-row ranges, fused wait and push); a separate instantiation so that the
// plain form keeps its 76-78 SGPRs
template <int kMultiTX, int kMultiTY, bool PEER = false>
__global__ __launch_bounds__(kMultiThreads) void d2q9_multi(const MultiArgs a) {
  constexpr int kMultiRX = kMultiTX + 2 * kMultiMaxT, kMultiRY = kMultiTY + 2 * kMultiMaxT;
  __shared__ float lds[2][9][kMultiRY * kMultiRX];
  __shared__ uint8_t lmask[kMultiRY * kMultiRX];
  __shared__ float wsum[kMultiMaxT][kMultiThreads / 64];
  const int tid = threadIdx.x;
  const int T = a.T;
  const int RX = kMultiTX + 2 * T, RY = kMultiTY + 2 * T;
  const int trow = blockIdx.x / a.tiles_x, tile_x = blockIdx.x - trow * a.tiles_x;
  int tile_y = trow < a.ty_split ? a.ty_begin + trow : a.ty_begin2 + (trow - a.ty_split);
  if constexpr (PEER)
    if (trow >= a.ty_split2) tile_y = a.ty_begin3 + (trow - a.ty_split2);
  const int gx0 = tile_x * kMultiTX - T, oy0 = tile_y * kMultiTY - T;  // region cell (0,0): column, owned-row index
  const bool periodic = (a.row_off == 0);
  const size_t ps = a.plane_stride;
  // stored row of region row ry: periodic wrap with one slab (kernels.cl:91-93), else the slab's halo rows
  auto stored_row = [&](int ry) {
    int r = oy0 + ry;
    if (periodic) {
      r %= a.rows;
      if (r < 0) r += a.rows;
      return r;
    }
    r += a.row_off;
    return r < 0 ? 0 : (r >= a.ext_rows ? a.ext_rows - 1 : r);  // clamped rows only feed cells that are never stored
  };

  // In-kernel wait for the neighbours' halo rows (halo_sync = 2).  No acquire fence behind it: L1 and L2 were invalidated
  // when this kernel started, the only loads of halo-row lines in this kernel are those of the edge tiles, and every
  // one of them comes after the poll below has seen the flag — which the producer raised after its write-through
  // stores had drained — so no cache can hold an older copy of those lines.
  if constexpr (PEER) {
    const MultiArgs *la = late_args<MultiArgs>();
    if ((la->peer_mode & 2) && (int)blockIdx.x < la->edge_blocks) {
      const HaloPeer *pp = la->peer;
      spin_on_flags(pp->wait_flags, la->wait_seq, pp->wait_err, pp->wait_ticks);
      __syncthreads();
    }
  }
  // region -> LDS (periodic wrap in x, kernels.cl:99-102)
  {
    const float inv = 1.0f / (float)RX;
    for (int i = tid; i < RX * RY; i += kMultiThreads) {
      const int ry = (int)(((float)i + 0.5f) * inv), rx = i - ry * RX;
      int gx = (gx0 + rx) % a.nx;
      if (gx < 0) gx += a.nx;
      const int sr = stored_row(ry);
      const float *p = a.src + (size_t)sr * a.row_stride + gx;
#pragma unroll
      for (int k = 0; k < 9; k++) lds[0][k][ry * kMultiRX + rx] = p[k * ps];
      lmask[ry * kMultiRX + rx] = a.mask[(size_t)sr * a.nx + gx];
    }
  }
  __syncthreads();

  for (int s = 1; s <= T; s++) {
    const int in = (s - 1) & 1, out = s & 1;
    const int w = RX - 2 * s, h = RY - 2 * s;
    const float inv = 1.0f / (float)w;
    const bool accel_step = (s < T) || a.accel_next;
    float sum = 0.f;
    for (int i = tid; i < w * h; i += kMultiThreads) {
      const int q = (int)(((float)i + 0.5f) * inv);
      const int rx = s + (i - q * w), ry = s + q;
      const int c = ry * kMultiRX + rx;
      float g[9], o[9];
      g[0] = lds[in][0][c];
      g[1] = lds[in][1][c - 1];
      g[2] = lds[in][2][c - kMultiRX];
      g[3] = lds[in][3][c + 1];
      g[4] = lds[in][4][c + kMultiRX];
      g[5] = lds[in][5][c - kMultiRX - 1];
      g[6] = lds[in][6][c - kMultiRX + 1];
      g[7] = lds[in][7][c + kMultiRX + 1];
      g[8] = lds[in][8][c + kMultiRX - 1];
      const bool obst = lmask[c] != 0;
      const float t = collide_cell(g, obst, a.omega, o);
      if (accel_step) {
        // by global row: with halo rows a slab may store a copy of row ny-2 besides (or instead of) its own
        int gy = (a.gy_off + stored_row(ry)) % a.ny_global;
        if (gy < 0) gy += a.ny_global;
        if (gy == a.ny_global - 2) accelerate_cell(o, obst, a.aw1, a.aw2);
      }
#pragma unroll
      for (int k = 0; k < 9; k++) lds[out][k][c] = o[k];
      // only the tile's own cells count (and, for tiles hanging over the grid edge, only real cells)
      const int ox = rx - T, oy = ry - T;
      if (ox >= 0 && ox < kMultiTX && oy >= 0 && oy < kMultiTY && tile_x * kMultiTX + ox < a.nx &&
          tile_y * kMultiTY + oy < a.rows)
        sum += t;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) wsum[s - 1][tid >> 6] = sum;
    __syncthreads();
  }

  // central tile -> global
  {
    const int fin = T & 1;
    for (int i = tid; i < kMultiTX * kMultiTY; i += kMultiThreads) {
      const int oy = i / kMultiTX, ox = i - oy * kMultiTX;
      const int gx = tile_x * kMultiTX + ox, orow = tile_y * kMultiTY + oy;
      if (gx < a.nx && orow < a.rows) {
        const int c = (oy + T) * kMultiRX + ox + T;
        float *d = a.dst + (size_t)(a.row_off + orow) * a.row_stride + gx;
#pragma unroll
        for (int k = 0; k < 9; k++) d[k * ps] = lds[fin][k][c];
      }
    }
  }
  if (tid < T) {
    float t = wsum[tid][0];
    for (int i = 1; i < kMultiThreads / 64; i++) t += wsum[tid][i];
    a.partials[(size_t)tid * a.partials_stride + blockIdx.x] = t;
  }
  // fused push (see MultiArgs): the edge tiles' share of the rows the ring neighbours need, 16 bytes per lane straight
  // from the LDS copy of the final state; a quad that hangs over the end of a row spills into the row padding
  // (plane_stride is a multiple of 64 floats >= nx: never read)
  if constexpr (PEER) {
    const MultiArgs *la = late_args<MultiArgs>();
    const int edge_blocks = la->edge_blocks;
    if ((la->peer_mode & 1) && (int)blockIdx.x < edge_blocks) {
      const HaloPeer *pp = la->peer;
      const int fin = T & 1;
      constexpr int Q = kMultiTX / 4;
      const int push_rows = pp->push_rows;
      float *const push_lo = pp->push[0][la->peer_buf & 1], *const push_hi = pp->push[1][la->peer_buf & 1];
      const int top0 = a.rows - push_rows;  // first owned row that goes north
      for (int i = tid; i < 9 * kMultiTY * Q; i += kMultiThreads) {
        const int k = i / (kMultiTY * Q), r = i - k * (kMultiTY * Q);
        const int oy = r / Q, q = r - oy * Q;
        const int orow = tile_y * kMultiTY + oy, gx = tile_x * kMultiTX + 4 * q;
        if (orow >= a.rows || gx >= a.nx) continue;
        const bool lo = orow < push_rows, hi = orow >= top0;
        if (!lo && !hi) continue;
        const float *src = &lds[fin][k][(oy + T) * kMultiRX + 4 * q + T];
        v4f v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
        if (lo) store4_through(push_lo + (size_t)orow * a.row_stride + k * ps + gx, v);
        if (hi) store4_through(push_hi + (size_t)(orow - top0) * a.row_stride + k * ps + gx, v);
      }
      release_pushed(pp->release);
      publish_when_last(pp->ticket, (unsigned)edge_blocks, pp->flag_lo, pp->flag_hi, la->seq, pp->release);
    }
  }
}

// ---- second reduction stage ------------------------------------------------------------------
// One workgroup per buffered step: sums that step's per-workgroup partials in a fixed order into
// av_sum[first + blockIdx.x] (double).  Replaces the reference's multi-pass reduce kernel
// (kernels.cl:234-290) and its in-place pass results.
static __global__ __launch_bounds__(kBlock) void reduce_partials(const float *partials, int stride, int nb, double *av_sum) {
  const float *p = partials + (size_t)blockIdx.x * stride;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += kBlock) acc += (double)p[i];
  __shared__ double wsum[kBlock / 64];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = wsum[0];
    for (int i = 1; i < kBlock / 64; i++) t += wsum[i];
    av_sum[blockIdx.x] = t;
  }
}

// ---- stand-alone accelerate_flow (kernels.cl:9-53): prologue of a run ---------------------------
static __global__ void accelerate_row(float *cells, unsigned long long plane_stride, unsigned long long row_stride,
                               const uint8_t *mask, int nx, int row, float aw1, float aw2) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= nx) return;
  const size_t c = (size_t)row * row_stride + x;
  float f3 = cells[3 * plane_stride + c], f6 = cells[6 * plane_stride + c], f7 = cells[7 * plane_stride + c];
  if (mask[(size_t)row * nx + x] == 0 && (f3 - aw1) > 0.0f && (f6 - aw2) > 0.0f && (f7 - aw2) > 0.0f) {
    cells[1 * plane_stride + c] += aw1;
    cells[5 * plane_stride + c] += aw2;
    cells[8 * plane_stride + c] += aw2;
    cells[3 * plane_stride + c] = f3 - aw1;
    cells[6 * plane_stride + c] = f6 - aw2;
    cells[7 * plane_stride + c] = f7 - aw2;
  }
}

// ---- initial state on the device (values of d2q9-bgk.c:529-550) ---------------------------------
static __global__ void init_cells(float *cells, unsigned long long plane_stride, unsigned long long row_stride, int nx,
                           size_t n, float w0, float w1, float w2) {
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    const size_t y = c / nx;
    const size_t i = y * row_stride + (c - y * nx);
    cells[i] = w0;
#pragma unroll
    for (int k = 1; k <= 4; k++) cells[k * plane_stride + i] = w1;
#pragma unroll
    for (int k = 5; k <= 8; k++) cells[k * plane_stride + i] = w2;
  }
}

// ---- read-back: row-interleaved device layout -> the reference's plane-major float[9][rows][nx] -------
// (into the grid that is not current — scratch between runs — so that the device-to-host transfer is nine large
// contiguous copies instead of 9*rows strided ones)
static __global__ void pack_planes(const float *cells, unsigned long long plane_stride, unsigned long long row_stride, int nx,
                            size_t n, float *out) {
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (size_t)gridDim.x * blockDim.x) {
    const size_t y = c / nx;
    const size_t i = y * row_stride + (c - y * nx);
#pragma unroll
    for (int k = 0; k < 9; k++) out[k * n + c] = cells[k * plane_stride + i];
  }
}

// ---- output stage: columns of final_state.dat + velocity sum (d2q9-bgk.c:787-832, 396-442) ------
static __global__ __launch_bounds__(kBlock) void final_fields(const float *cells, unsigned long long plane_stride,
                                                       unsigned long long row_stride, int nx, const uint8_t *mask,
                                                       size_t n, float density, float *u_x, float *u_y, float *u,
                                                       float *pressure, float *partials) {
  const float c_sq = 1.0f / 3.0f;
  float tot_u = 0.0f;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    float ux = 0.0f, uy = 0.0f, uu = 0.0f, pr = density * c_sq;
    if (mask[i] == 0) {
      float f[9];
      float local_density = 0.0f;
      const size_t y = i / nx;
      const size_t cell = y * row_stride + (i - y * nx);
#pragma unroll
      for (int k = 0; k < 9; k++) {
        f[k] = cells[k * plane_stride + cell];
        local_density += f[k];
      }
      ux = (f[1] + f[5] + f[8] - f[3] - f[6] - f[7]) / local_density;
      uy = (f[2] + f[5] + f[6] - f[4] - f[7] - f[8]) / local_density;
      uu = sqrtf(ux * ux + uy * uy);
      pr = local_density * c_sq;
      tot_u += uu;
    }
    if (u_x) u_x[i] = ux;
    if (u_y) u_y[i] = uy;
    if (u) u[i] = uu;
    if (pressure) pressure[i] = pr;
  }
  block_store_partial(tot_u, partials);
}

// ---- roofline denominator: float4 streaming copy ---------------------------------------------
// VALU roofline calibration: eight independent chains of packed fp32 fused multiply-adds per thread — the instruction the
// deep window kernel's collision is made of — so that the issue rate, not a dependency, bounds the loop.
static __global__ __launch_bounds__(kBlock) void valu_spin(float *out, int iters, float seed) {
  v2f acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = splat2(seed + 0.125f * (float)i + 1e-3f * (float)(threadIdx.x & 7));
  const v2f a = splat2(0.99990f), b = splat2(1.0e-4f);
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = fma2(acc[i], a, b);
    }
  }
  v2f s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; i++) s += acc[i];
  out[(size_t)blockIdx.x * kBlock + threadIdx.x] = s.x + s.y;
}

static __global__ __launch_bounds__(kBlock) void copy_f4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) out[i] = in[i];
}

// the same with the access pattern that proved best for the step kernels: one 4-KiB tile per workgroup,
// non-temporal loads and stores
static __global__ __launch_bounds__(kBlock) void copy_f4_nt(const float *__restrict__ in, float *__restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n4) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(in) + i);
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(out) + i);
  }
}

}  // namespace lbm
