#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { v2f r = {x, x}; return r; }
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ v2f collide_pair(const v2f (&g)[9], bool oa, bool ob, float omega, v2f (&out)[9]) {
#pragma clang fp contract(off)
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;
  v2f dens = g[0] + g[1];
  dens += g[2]; dens += g[3]; dens += g[4]; dens += g[5]; dens += g[6]; dens += g[7]; dens += g[8];
  v2f densinv = {__builtin_amdgcn_rcpf(dens.x), __builtin_amdgcn_rcpf(dens.y)};
  const v2f da = g[5] - g[7], db = g[8] - g[6];
  const v2f jx = (g[1] - g[3]) + (da + db);
  const v2f jy = (g[2] - g[4]) + (da - db);
  const v2f usq = fma2(jx, jx, jy * jy);
  const v2f h = splat(1.5f) * densinv;
  const v2f c = fma2(-h, usq, dens);
  const v2f h3 = splat(3.0f) * h;
  const v2f jp = jx + jy, jm = jx - jy;
  const v2f ax = fma2(h3 * jx, jx, c), ay = fma2(h3 * jy, jy, c), ap = fma2(h3 * jp, jp, c), am = fma2(h3 * jm, jm, c);
  v2f eq[9];
  eq[0] = splat(w0) * c;
  eq[1] = splat(w1) * fma2(splat(3.0f), jx, ax); eq[3] = splat(w1) * fma2(splat(-3.0f), jx, ax);
  eq[2] = splat(w1) * fma2(splat(3.0f), jy, ay); eq[4] = splat(w1) * fma2(splat(-3.0f), jy, ay);
  eq[5] = splat(w2) * fma2(splat(3.0f), jp, ap); eq[7] = splat(w2) * fma2(splat(-3.0f), jp, ap);
  eq[8] = splat(w2) * fma2(splat(3.0f), jm, am); eq[6] = splat(w2) * fma2(splat(-3.0f), jm, am);
  const int opp[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const v2f r = fma2(splat(omega), eq[k] - g[k], g[k]);
    out[k].x = oa ? g[opp[k]].x : r.x;
    out[k].y = ob ? g[opp[k]].y : r.y;
  }
  v2f u = {__builtin_amdgcn_sqrtf(usq.x) * densinv.x, __builtin_amdgcn_sqrtf(usq.y) * densinv.y};
  u.x = oa ? 0.f : u.x; u.y = ob ? 0.f : u.y;
  return u;
}
__device__ __forceinline__ void accelerate_pair(v2f (&f)[9], bool oa, bool ob, float aw1, float aw2) {
#pragma clang fp contract(off)
  const bool ta = !oa && (f[3].x - aw1) > 0.f && (f[6].x - aw2) > 0.f && (f[7].x - aw2) > 0.f;
  const bool tb = !ob && (f[3].y - aw1) > 0.f && (f[6].y - aw2) > 0.f && (f[7].y - aw2) > 0.f;
  if (ta) { f[1].x += aw1; f[5].x += aw2; f[8].x += aw2; f[3].x -= aw1; f[6].x -= aw2; f[7].x -= aw2; }
  if (tb) { f[1].y += aw1; f[5].y += aw2; f[8].y += aw2; f[3].y -= aw1; f[6].y -= aw2; f[7].y -= aw2; }
}
__device__ __forceinline__ float dpp_below(float v, float old) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_above(float v, float old) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xf, 0xf, false)); }
__device__ __forceinline__ v2f from_west(v2f p, float halo) { v2f r = {dpp_below(p.y, halo), p.x}; return r; }
__device__ __forceinline__ v2f from_east(v2f p, float halo) { v2f r = {p.y, dpp_above(p.x, halo)}; return r; }
__device__ __forceinline__ float dpp_below0(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_above0(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ v2f from_west(v2f p) { v2f r = {dpp_below0(p.y), p.x}; return r; }
__device__ __forceinline__ v2f from_east(v2f p) { v2f r = {p.y, dpp_above0(p.x)}; return r; }

struct DeepArgs {
  const float *src; float *dst; const uint8_t *mask;
  unsigned long long plane_stride, row_stride;
  int nx, ny, strips, lanes_out, accel_row, accel_next;
  float omega, aw1, aw2;
  const int *chunk_start;
  float *partials;  // [D][units]
  int units, nbands, units_per_band;
};

struct PairLoads { v2f c[9]; float h0, h1, h2; uint32_t m; };

__device__ __forceinline__ void issue_pair_loads(const DeepArgs &a, int r, int xcol, int xhw, int xhe, int lane, PairLoads &in) {
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int r_s = (r == 0) ? a.ny - 1 : r - 1, r_n = (r == a.ny - 1) ? 0 : r + 1;
  const float *Rc = a.src + (size_t)r * rs, *Rs = a.src + (size_t)r_s * rs, *Rn = a.src + (size_t)r_n * rs;
  auto ld = [&](const float *p) { return *reinterpret_cast<const v2f *>(p + xcol); };
  in.c[0] = ld(Rc); in.c[1] = ld(Rc + ps); in.c[3] = ld(Rc + 3 * ps);
  in.c[2] = ld(Rs + 2 * ps); in.c[5] = ld(Rs + 5 * ps); in.c[6] = ld(Rs + 6 * ps);
  in.c[4] = ld(Rn + 4 * ps); in.c[7] = ld(Rn + 7 * ps); in.c[8] = ld(Rn + 8 * ps);
  in.m = *reinterpret_cast<const uint16_t *>(a.mask + (size_t)r * a.nx + xcol);
  in.h0 = in.h1 = in.h2 = 0.f;
  if (lane == 0 || lane == 63) {
    const bool lo = lane == 0;
    in.h0 = lo ? Rc[ps + xhw] : Rc[3 * ps + xhe];
    in.h1 = lo ? Rs[5 * ps + xhw] : Rs[6 * ps + xhe];
    in.h2 = lo ? Rn[8 * ps + xhw] : Rn[7 * ps + xhe];
  }
}
__device__ __forceinline__ float collide2(const v2f (&g)[9], uint32_t m, float omega, bool accel_uniform, float aw1, float aw2, v2f (&o)[9]) {
  const bool oa = (m & 0xffu) != 0, ob = (m & 0xff00u) != 0;
  const v2f u = collide_pair(g, oa, ob, omega, o);
  if (accel_uniform) accelerate_pair(o, oa, ob, aw1, aw2);   // a scalar branch: the row is the same for the whole wave
  return u.x + u.y;
}

struct PairWindow { v2f mid[3]; v2f S0[3], S1[3]; };
constexpr int kSlotFloats = 128;          // one slot: 64 lanes x 2 floats
constexpr int kWinFloats = 9 * kSlotFloats;
__device__ __forceinline__ v2f lds_pair(const float *p) { return *reinterpret_cast<const v2f *>(p); }
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
__device__ __forceinline__ v2f lds_pair_u(const float *p) { const f2u t = *reinterpret_cast<const f2u *>(p); v2f r = {t.x, t.y}; return r; }

template <int D, int WL, bool UP, int NBUF, bool NT>
__device__ __forceinline__ void deep_sweep(const DeepArgs &a, float *lds, int ys, int ye, int xcol, int xhw, int xhe, int lane, bool owner, int unit) {
  const size_t ps = a.plane_stride;
  auto wrap = [&](int r) { return r < 0 ? r + a.ny : (r >= a.ny ? r - a.ny : r); };
  const int n = ye - ys, d = UP ? 1 : -1;
  const int r0 = UP ? ys - (D - 1) : ye + (D - 2);
  const int last = n + 2 * (D - 1) - 1;
  float sum[D];
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
    for (int i = 0; i < 3; i++) w[l].mid[i] = w[l].S0[i] = w[l].S1[i] = splat(0.f);
  }
  float *const lw = lds + 2 + 2 * lane;   // +2 floats: lane 0 reads one float below its slot
  // Level l works on the row level 0 worked on l iterations earlier: what is known about a row (is it the accelerated
  // row, is it one of this chunk's own rows) travels in scalar shift registers, bit l = level l of this iteration.
  uint32_t accbits = 0, ownbits = 0;
  const uint32_t accmask = a.accel_next ? 0xffffffffu : ~(1u << (D - 1));
  PairLoads inA, inB;
  issue_pair_loads(a, wrap(r0), xcol, xhw, xhe, lane, inA);
  if (NBUF == 2) issue_pair_loads(a, wrap(r0 + d), xcol, xhw, xhe, lane, inB);
  // the six window planes of level l (1..D-1): unshifted / from west / from east of the middle row and of the trail row
  auto window_read = [&](int l, auto parc, v2f (&q)[6]) __attribute__((always_inline)) {
    constexpr int par = decltype(parc)::value;
    if ((l - 1) < WL) {
      const float *W = lw + (l - 1) * kWinFloats;
      const float *Wp = W + (3 + 3 * par) * kSlotFloats;
      q[0] = lds_pair(W); q[1] = lds_pair_u(W + kSlotFloats - 1); q[2] = lds_pair_u(W + 2 * kSlotFloats + 1);
      q[3] = lds_pair(Wp); q[4] = lds_pair_u(Wp + kSlotFloats - 1); q[5] = lds_pair_u(Wp + 2 * kSlotFloats + 1);
    } else {
      const PairWindow &R = w[(l - 1) < WL ? 0 : (l - 1 - WL)];
      q[0] = R.mid[0]; q[1] = from_west(R.mid[1]); q[2] = from_east(R.mid[2]);
      q[3] = R.S0[0]; q[4] = from_west(R.S0[1]); q[5] = from_east(R.S0[2]);
    }
  };
  auto iterate = [&](const int k, auto parc, PairLoads &in) __attribute__((always_inline)) {
    constexpr int par = decltype(parc)::value;
    v2f top[9];
    uint32_t m_top;
    v2f pre[6];
    const int row0 = wrap(r0 + k * d);
    accbits = ((accbits << 1) | ((row0 == a.accel_row) ? 1u : 0u)) & accmask;
    ownbits = (ownbits << 1) | ((k >= D - 1 && k <= n + D - 2) ? 1u : 0u);
    if (k >= 2) window_read(1, parc, pre);
    {  // level 0 from the loaded source rows
      v2f g[9];
      g[0] = in.c[0]; g[2] = in.c[2]; g[4] = in.c[4];
      g[1] = from_west(in.c[1], in.h0); g[5] = from_west(in.c[5], in.h1); g[8] = from_west(in.c[8], in.h2);
      g[3] = from_east(in.c[3], in.h0); g[6] = from_east(in.c[6], in.h1); g[7] = from_east(in.c[7], in.h2);
      const float t = collide2(g, in.m, a.omega, (accbits & 1u) != 0, a.aw1, a.aw2, top);
      m_top = in.m;
      if ((ownbits & 1u) && owner) sum[0] += t;
      if (k + NBUF <= last) issue_pair_loads(a, wrap(r0 + (k + NBUF) * d), xcol, xhw, xhe, lane, in);
    }
#pragma unroll
    for (int l = 1; l < D; l++) {
      v2f nxt[9];
      uint32_t m_nxt = 0;
      const bool active = k >= 2 * l;
      const bool in_lds = (l - 1) < WL;
      float *const W = lw + (l - 1) * kWinFloats;
      float *const Wp = W + (3 + 3 * par) * kSlotFloats;
      PairWindow &R = w[in_lds ? 0 : (l - 1 - WL)];
      if (active) {
        v2f q[6];
#pragma unroll
        for (int i = 0; i < 6; i++) q[i] = pre[i];
        if (l + 1 < D && k >= 2 * (l + 1)) window_read(l + 1, parc, pre);   // the next level's window, early
        v2f g[9];
        g[0] = q[0]; g[1] = q[1]; g[3] = q[2];
        if (UP) {
          g[2] = q[3]; g[5] = q[4]; g[6] = q[5];
          g[4] = top[4]; g[8] = from_west(top[8]); g[7] = from_east(top[7]);
        } else {
          g[4] = q[3]; g[8] = q[4]; g[7] = q[5];
          g[2] = top[2]; g[5] = from_west(top[5]); g[6] = from_east(top[6]);
        }
        m_nxt = m_mid[l - 1];
        const float t = collide2(g, m_nxt, a.omega, ((accbits >> l) & 1u) != 0, a.aw1, a.aw2, nxt);
        if (l < D - 1) { if (((ownbits >> l) & 1u) && owner) sum[l] += t; }
        else if (owner) {
          sum[l] += t;
          const int y = r0 + (k - l) * d;
          float *dp = a.dst + (size_t)y * a.row_stride + xcol;
#pragma unroll
          for (int kk = 0; kk < 9; kk++) { if (NT) __builtin_nontemporal_store(nxt[kk], reinterpret_cast<v2f *>(dp + kk * ps)); else *reinterpret_cast<v2f *>(dp + kk * ps) = nxt[kk]; }
        }
      }
      if (in_lds) {
        *reinterpret_cast<v2f *>(W) = top[0]; *reinterpret_cast<v2f *>(W + kSlotFloats) = top[1]; *reinterpret_cast<v2f *>(W + 2 * kSlotFloats) = top[3];
        *reinterpret_cast<v2f *>(Wp) = UP ? top[2] : top[4];
        *reinterpret_cast<v2f *>(Wp + kSlotFloats) = UP ? top[5] : top[8];
        *reinterpret_cast<v2f *>(Wp + 2 * kSlotFloats) = UP ? top[6] : top[7];
      } else {
        R.mid[0] = top[0]; R.mid[1] = top[1]; R.mid[2] = top[3];
#pragma unroll
        for (int i = 0; i < 3; i++) R.S0[i] = R.S1[i];
        R.S1[0] = UP ? top[2] : top[4]; R.S1[1] = UP ? top[5] : top[8]; R.S1[2] = UP ? top[6] : top[7];
      }
      m_mid[l - 1] = m_top;
      if (!active) break;
      if (l < D - 1) {
#pragma unroll
        for (int kk = 0; kk < 9; kk++) top[kk] = nxt[kk];
        m_top = m_nxt;
      }
    }
  };
  for (int k = 0; k <= last; k += 2) {
    iterate(k, std::integral_constant<int, 0>(), inA);
    if (k + 1 <= last) iterate(k + 1, std::integral_constant<int, 1>(), NBUF == 2 ? inB : inA);
  }
#pragma unroll
  for (int l = 0; l < D; l++) {
    const float s = wave_sum(sum[l]);
    if (lane == 0) a.partials[(size_t)l * a.units + unit] = s;
  }
}

template <int D, int WL, int NBUF, bool NT>
__global__ __launch_bounds__(64, 2) void d2q9_deep(const DeepArgs a) {
  __shared__ float lds[WL * kWinFloats + 4];
  const int lane = threadIdx.x;
  const int band = blockIdx.x % a.nbands, slot = blockIdx.x / a.nbands;
  if (slot >= a.units_per_band) return;
  const int unit = band * a.units_per_band + slot;
  if (unit >= a.units) return;
  const int chunk = unit / a.strips, strip = unit - chunk * a.strips;
  const int ys = a.chunk_start[chunk], ye = a.chunk_start[chunk + 1];
  constexpr int HL = D / 2;  // halo lanes per side: level L loses L-1 cells per side
  const int q2 = a.nx >> 1;
  const int qcol = strip * a.lanes_out + lane - HL;
  const bool owner = lane >= HL && lane < HL + a.lanes_out && qcol < q2;
  int qw = qcol % q2; if (qw < 0) qw += q2;
  const int xcol = qw * 2;
  const int xhw = xcol == 0 ? a.nx - 1 : xcol - 1, xhe = xcol + 2 >= a.nx ? 0 : xcol + 2;
  if (ys >= ye) return;
  if (__builtin_amdgcn_readfirstlane((int)((chunk & 1) == 0))) deep_sweep<D, WL, true, NBUF, NT>(a, lds, ys, ye, xcol, xhw, xhe, lane, owner, unit);
  else deep_sweep<D, WL, false, NBUF, NT>(a, lds, ys, ye, xcol, xhw, xhe, lane, owner, unit);
}

// ---- harness: naive single-step kernel (independent scalar arithmetic) + timing ----
__device__ __forceinline__ float collide_scalar(const float (&g)[9], bool obstacle, float omega, float (&out)[9]) {
#pragma clang fp contract(off)
  const float w0 = 4.0f / 9.0f, w1 = 1.0f / 9.0f, w2 = 1.0f / 36.0f;
  float dens = g[0] + g[1];
  dens += g[2]; dens += g[3]; dens += g[4]; dens += g[5]; dens += g[6]; dens += g[7]; dens += g[8];
  const float densinv = __builtin_amdgcn_rcpf(dens);
  const float da = g[5] - g[7], db = g[8] - g[6];
  const float jx = (g[1] - g[3]) + (da + db);
  const float jy = (g[2] - g[4]) + (da - db);
  const float usq = __builtin_fmaf(jx, jx, jy * jy);
  const float h = 1.5f * densinv;
  const float c = __builtin_fmaf(-h, usq, dens);
  const float h3 = 3.0f * h;
  const float jp = jx + jy, jm = jx - jy;
  const float ax = __builtin_fmaf(h3 * jx, jx, c), ay = __builtin_fmaf(h3 * jy, jy, c);
  const float ap = __builtin_fmaf(h3 * jp, jp, c), am = __builtin_fmaf(h3 * jm, jm, c);
  float eq[9];
  eq[0] = w0 * c;
  eq[1] = w1 * __builtin_fmaf(3.0f, jx, ax); eq[3] = w1 * __builtin_fmaf(-3.0f, jx, ax);
  eq[2] = w1 * __builtin_fmaf(3.0f, jy, ay); eq[4] = w1 * __builtin_fmaf(-3.0f, jy, ay);
  eq[5] = w2 * __builtin_fmaf(3.0f, jp, ap); eq[7] = w2 * __builtin_fmaf(-3.0f, jp, ap);
  eq[8] = w2 * __builtin_fmaf(3.0f, jm, am); eq[6] = w2 * __builtin_fmaf(-3.0f, jm, am);
  const int opp[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
  for (int k = 0; k < 9; k++) out[k] = obstacle ? g[opp[k]] : __builtin_fmaf(omega, eq[k] - g[k], g[k]);
  return obstacle ? 0.f : __builtin_amdgcn_sqrtf(usq) * densinv;
}
__global__ void naive_step(const DeepArgs a) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= a.nx) return;
  const size_t ps = a.plane_stride, rs = a.row_stride;
  const int xw = x == 0 ? a.nx - 1 : x - 1, xe = x == a.nx - 1 ? 0 : x + 1;
  const int ysr = y == 0 ? a.ny - 1 : y - 1, yn = y == a.ny - 1 ? 0 : y + 1;
  float g[9], o[9];
  g[0] = a.src[y * rs + x]; g[1] = a.src[y * rs + ps + xw]; g[3] = a.src[y * rs + 3 * ps + xe];
  g[2] = a.src[ysr * rs + 2 * ps + x]; g[5] = a.src[ysr * rs + 5 * ps + xw]; g[6] = a.src[ysr * rs + 6 * ps + xe];
  g[4] = a.src[yn * rs + 4 * ps + x]; g[8] = a.src[yn * rs + 8 * ps + xw]; g[7] = a.src[yn * rs + 7 * ps + xe];
  const bool ob = a.mask[(size_t)y * a.nx + x] != 0;
  collide_scalar(g, ob, a.omega, o);
  if (y == a.accel_row && !ob && (o[3] - a.aw1) > 0.f && (o[6] - a.aw2) > 0.f && (o[7] - a.aw2) > 0.f) {
    o[1] += a.aw1; o[5] += a.aw2; o[8] += a.aw2; o[3] -= a.aw1; o[6] -= a.aw2; o[7] -= a.aw2;
  }
  for (int k = 0; k < 9; k++) a.dst[y * rs + k * ps + x] = o[k];
}
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
template <int D, int WL>
static void launch_deep(const DeepArgs &a, int grid, int nbuf, int nt) {
  if (nbuf == 2 && nt) hipLaunchKernelGGL((d2q9_deep<D, WL, 2, true>), dim3(grid), dim3(64), 0, 0, a);
  else if (nbuf == 2) hipLaunchKernelGGL((d2q9_deep<D, WL, 2, false>), dim3(grid), dim3(64), 0, 0, a);
  else if (nt) hipLaunchKernelGGL((d2q9_deep<D, WL, 1, true>), dim3(grid), dim3(64), 0, 0, a);
  else hipLaunchKernelGGL((d2q9_deep<D, WL, 1, false>), dim3(grid), dim3(64), 0, 0, a);
}
int main(int argc, char **argv) {
  const int nx = argc > 1 ? atoi(argv[1]) : 1024, ny = argc > 2 ? atoi(argv[2]) : 1024;
  const int D = argc > 3 ? atoi(argv[3]) : 6, WL = argc > 4 ? atoi(argv[4]) : 4;
  const int chunk_rows = argc > 5 ? atoi(argv[5]) : 64, iters = argc > 6 ? atoi(argv[6]) : 20, check = argc > 7 ? atoi(argv[7]) : 1;
  const int nbuf = argc > 8 ? atoi(argv[8]) : 1, nt = argc > 9 ? atoi(argv[9]) : 0;
  const size_t ps = nx, rs = 9 * (size_t)nx, total = rs * ny;
  std::vector<float> h(total);
  std::vector<uint8_t> m((size_t)nx * ny, 0);
  srand(1);
  const float w[9] = {4.f / 9, 1.f / 9, 1.f / 9, 1.f / 9, 1.f / 9, 1.f / 36, 1.f / 36, 1.f / 36, 1.f / 36};
  for (int y = 0; y < ny; y++)
    for (int k = 0; k < 9; k++)
      for (int x = 0; x < nx; x++) h[y * rs + k * ps + x] = 0.1f * w[k] * (1.f + 0.2f * ((rand() & 1023) / 1024.f - 0.5f));
  for (int y = 0; y < ny; y++)
    for (int x = 0; x < nx; x++) m[(size_t)y * nx + x] = (x == 0 || y == 0 || x == nx - 1 || y == ny - 1 || (rand() % 97) == 0) ? 1 : 0;
  float *d0, *d1, *d2, *d3, *parts; uint8_t *dm; int *dcs;
  CK(hipMalloc(&d0, total * 4)); CK(hipMalloc(&d1, total * 4)); CK(hipMalloc(&d2, total * 4)); CK(hipMalloc(&d3, total * 4));
  CK(hipMalloc(&dm, m.size()));
  CK(hipMemcpy(d0, h.data(), total * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, h.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dm, m.data(), m.size(), hipMemcpyHostToDevice));
  const int HL = D / 2, lanes_out = 64 - 2 * HL;
  const int strips = (nx / 2 + lanes_out - 1) / lanes_out;
  std::vector<int> cs;
  for (int y = 0; y < ny; y += chunk_rows) cs.push_back(y);
  cs.push_back(ny);
  const int chunks = (int)cs.size() - 1, units = chunks * strips;
  CK(hipMalloc(&dcs, cs.size() * 4)); CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&parts, (size_t)units * 8 * 4));
  DeepArgs a{};
  a.src = d0; a.dst = d1; a.mask = dm; a.plane_stride = ps; a.row_stride = rs; a.nx = nx; a.ny = ny; a.strips = strips; a.lanes_out = lanes_out;
  a.accel_row = ny - 2; a.accel_next = 1; a.omega = 1.85f; a.aw1 = 0.1f * 0.005f / 9.f; a.aw2 = 0.1f * 0.005f / 36.f;
  a.chunk_start = dcs; a.partials = parts; a.units = units; a.nbands = 8; a.units_per_band = (units + 7) / 8;
  const int grid = a.units_per_band * 8;
  auto launch = [&](const DeepArgs &aa) {
    if (D == 4 && WL == 3) launch_deep<4, 3>(aa, grid, nbuf, nt);
    else if (D == 5 && WL == 4) launch_deep<5, 4>(aa, grid, nbuf, nt);
    else if (D == 6 && WL == 4) launch_deep<6, 4>(aa, grid, nbuf, nt);
    else if (D == 6 && WL == 3) launch_deep<6, 3>(aa, grid, nbuf, nt);
    else { fprintf(stderr, "no such instantiation\n"); exit(2); }
  };
  if (check) {
    launch(a);
    DeepArgs b = a;
    float *s = d2, *t = d3;
    for (int i = 0; i < D; i++) { b.src = s; b.dst = t; hipLaunchKernelGGL(naive_step, dim3((nx + 255) / 256, ny), dim3(256), 0, 0, b); float *u = s; s = t; t = u; }
    CK(hipDeviceSynchronize());
    std::vector<float> r1(total), r2(total);
    CK(hipMemcpy(r1.data(), d1, total * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), s, total * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < total; i++) if (memcmp(&r1[i], &r2[i], 4)) { if (!bad) first = i; bad++; }
    printf("check %dx%d D=%d WL=%d: %zu of %zu values differ", nx, ny, D, WL, bad, total);
    if (bad) printf(" (first at y=%zu k=%zu x=%zu: %g vs %g)", first / rs, (first % rs) / ps, first % ps, r1[first], r2[first]);
    printf("\n");
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    DeepArgs b = a;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) { launch(b); float *u = (float *)b.src; b.src = b.dst; b.dst = u; }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%dx%d D=%d WL=%d chunk %d nbuf %d nt %d units %d: %.1f us/launch  %.1f GLUPS\n", nx, ny, D, WL, chunk_rows, nbuf, nt, units, ms / iters * 1e3, (double)nx * ny * D * iters / ms / 1e6);
  }
  return 0;
}
