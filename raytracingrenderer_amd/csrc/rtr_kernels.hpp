// The path tracer's kernels (gfx950, wave64).
//
// One render = ray generation -> primary trace -> classify, the sampling loop, and
// a final resolve.  The sampling loop exists twice, with the same image bit for bit:
//
//   fused      k_render_fused: one persistent kernel, one path per lane, path state
//              in registers, both rays of a vertex traced by the lane itself; for
//              scenes small enough to be staged in LDS (and small trees in general).
//   wavefront  a loop of k_logic (consume last iteration's hits, Russian roulette
//              result, path regeneration, light sampling, BSDF sampling, ray emission
//              with 64-wide ballot compaction into the ray queue) and a traversal kernel
//              (closest hit for every queued ray: k_trace_queue for scenes in LDS,
//              k_trace_queue_stream for trees read from memory); path state lives in HBM
//              as float4 structure-of-arrays indexed by slot.
//
// A work unit is (pixel, chunk of consecutive samples).  A lane or slot sums its
// unit's samples in order and writes ONE partial sum per unit, so the image needs
// no atomics and is bit-reproducible: it does not depend on the loop variant, the
// pool size, the scheduling order or how columns are split across GPUs.
#pragma once
#include "rtr_shade.hpp"

namespace rtr {

constexpr int kBlock = 256;
constexpr int kLogicBlock = 256;
constexpr int kLogicWaves = kLogicBlock / 64;
constexpr int kSegment = 2 * kLogicBlock;  // queue entries owned by one k_logic workgroup
// The traversal kernel reads kGroup consecutive segments as one packed stream, so only the
// last wave per GROUP is partly filled; the path-logic workgroups can then be small (better
// latency hiding, several per CU) without leaving half a wave idle per workgroup.
constexpr int kGroup = 4;
constexpr int kStreamParts = kGroup * kLogicBlock / kBlock;  // traversal workgroups per stream
constexpr int kGroupParts = 2 * kStreamParts;  // a group has a shadow stream and an extension stream

struct CameraSetup {  // src/Render.cpp:639-649, computed on the host
  V3 eye, front, hstep, vstep;
};

struct PixelCache {  // first path vertex of every pixel: identical for all its samples
  float4 *pos_mat;   // pos.xyz, bits(material)      (material = -1: miss, -2: emissive)
  float4 *nrm;       // unit normal
  float4 *wi;        // -unit(primary dir), w = bits(global pixel id i*H+j)
};

struct PathPool {
  float4 *A;   // vertex pos.xyz, bits(material)
  float4 *B;   // unit normal N.xyz, bits(depth)
  float4 *C;   // wi.xyz, bits(local pixel)
  float4 *D;   // beta.xyz, bits(sample k)
  float4 *E;   // L.xyz (radiance of the current path), bits(k_end)
  float4 *F;   // acc.xyz (sum over the unit's finished samples), bits(chunk)
  float4 *G;   // pending next-event weight Le*f*cos*cos, light pdf
  float4 *Hh;  // light point.xyz, bits(flags)
  float4 *I;   // beta after the pending bounce, bits(global pixel id)
  float4 *J;   // shadow ray direction
  float4 *K;   // extension ray direction
  float4 *hs;   // shadow hit: {hit point.xyz, distance}; point.x = NaN for a miss
  float4 *he;   // extension hit: {hit point.xyz, bits(prim id or -1)}
  float4 *he2;  // extension hit: {unit normal.xyz, -}
};

enum : uint32_t { kLive = 1u, kHasShadow = 2u, kHasExt = 4u, kHasUnit = 8u };

struct Counters {
  unsigned int n_shade;  // pixels that need sampling (k_classify)
  unsigned int pad;
  unsigned long long vertices, rays;  // totals of the fused sampling kernel
  unsigned long long next_unit;       // its cursor: first unit no wave has reserved yet
  // cursors of k_trace_queue_stream over the (workgroup, ray kind) queue segments of a
  // launch, one per half of the pool; k_logic resets the one of its half
  unsigned int trace_next[2];
  // rays queued by the k_logic launch of [half of the pool][iteration parity]: lets the
  // traversal kernel see how full the queue segments of its launch are
  unsigned int queued[2][2];
  // work of the streaming traversal launches of a render: box tests, primitive tests
  unsigned long long boxes, prims;
};

// The wavefront loop's only global atomic is the unit cursor (one add per 64 units).
// Every k_logic workgroup owns one record and one fixed segment of the ray queue;
// the host adds the records up.  (Per-wave, and then per-workgroup, atomics on
// shared queue counters used to be the largest single cost of the path-logic kernel.)
struct BlockCounters {
  unsigned int qcount[2][2];    // rays this workgroup queued: [iteration parity][0 shadow, 1 extension]
  unsigned int done;            // work units finished so far
  unsigned int pad;
  unsigned long long vertices;  // path vertices shaded so far
  unsigned long long rays[2];   // rays traced so far, per stream (written by the traversal kernel)
};

struct RenderArgs {
  int32_t W, H, spp, n_cols, n_pix;  // n_pix = n_cols * H (local pixels)
  uint32_t seed;
  int32_t max_depth;
  float rr;
  int32_t chunk_spp, n_chunks, pool;
  uint32_t n_shade, n_units;  // wavefront loop only: read back after k_classify
  int32_t first;              // 1 on the first iteration: no slot has a unit yet
};

RTR_DEV unsigned lane_id() { return threadIdx.x & 63u; }
RTR_DEV unsigned long long lanes_below() { return (1ull << lane_id()) - 1ull; }

// Which columns of the image a render covers: the image is cut into tiles of tile_cols
// columns and partition part_index of part_count takes the tiles t with
// t % part_count == part_index, packed in increasing column order.
struct ColumnMap {
  int32_t part_index, part_count, tile_cols;
  // local column -> image column (only the last tile of the image can be narrower than
  // tile_cols, so every tile before a given local column is full)
  RTR_DEV int column(int lc) const {
    if (part_count <= 1) return lc;
    return ((lc / tile_cols) * part_count + part_index) * tile_cols + lc % tile_cols;
  }
};

// ---- ray generation: src/Render.cpp:654 ----
__global__ void __launch_bounds__(kBlock)
k_raygen(RenderArgs a, CameraSetup cam, ColumnMap cm, float4 *__restrict__ dir) {
  int lp = blockIdx.x * kBlock + threadIdx.x;
  if (lp >= a.n_pix) return;
  int lc = lp / a.H, j = lp - lc * a.H;
  int i = cm.column(lc);
  V3 d = unit(cam.front + (float)(-a.W / 2 + i) * cam.hstep + (float)(-a.H / 2 + j) * cam.vstep);
  dir[lp] = pack(d, (uint32_t)(i * a.H + j));
}

// ---- the framebuffers of all partitions, one after the other -> the whole image ----
// Partition r's columns (n_cols of them, column-major, 3 floats per pixel) start at
// float offset part_off[r] of `parts`; the image has the reference's layout (i*H + j).
constexpr int kMaxParts = 64;
struct PartOffsets {
  unsigned long long off[kMaxParts];
};
__global__ void __launch_bounds__(kBlock)
k_unpartition(int W, int H, int part_count, int tile_cols, PartOffsets po,
              const float *__restrict__ parts, float *__restrict__ image) {
  const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;  // pixel i*H + j
  if (idx >= (size_t)W * H) return;
  const int i = (int)(idx / H), j = (int)(idx - (size_t)i * H);
  const int tile = i / tile_cols;
  const int r = tile % part_count;
  const int lc = (tile / part_count) * tile_cols + i % tile_cols;
  const float *src = parts + po.off[r] + 3 * ((size_t)lc * H + j);
  float *dst = image + 3 * idx;
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

// ---- traversal over explicit ray arrays (the rtr_trace_closest entry point) ----
// k_trace_api, k_check_connectable and k_trace_queue are launched for scenes staged in LDS
// (kLds = true); trees read from memory take the *_stream kernels further down, which
// keep lanes busy across rays.  k_trace_primary and k_render_fused run in both flavours.
// Closest hit for the 64 rays of a wave, from staged (LDS) or global geometry;
// kLds is chosen by the host from the scene's size.  Every lane of the wave must
// call this (valid = false where there is no ray).
// Per-wave LDS of the two traversal flavours: small scenes only need the stack of
// the literal merge-tree fallback.
struct SmallShared { MergeStack merge; };
template <bool kLds> struct WaveLds { typedef WaveShared type; };
template <> struct WaveLds<true> { typedef SmallShared type; };

template <bool kLds>
RTR_DEV HitRec trace_any(const DeviceScene &sc, const LdsScene &lds, bool valid, V3 o, V3 d, V3 dn,
                         typename WaveLds<kLds>::type &ws) {
  if constexpr (kLds) {
    return trace_all(sc, lds, valid, o, d, dn, ws.merge);
  } else {
    return trace_wave(global_geom(sc), sc.n_nodes, sc.fast_planes != 0, sc.nested_boxes != 0, valid, o, d, dn,
                      ws);
  }
}

template <bool kLds>
__global__ void __launch_bounds__(kBlock)
k_trace_api(DeviceScene sc, const float *__restrict__ org, const float *__restrict__ dir, size_t n,
            int32_t *__restrict__ prim, float *__restrict__ t, float *__restrict__ pos,
            float *__restrict__ nrm) {
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  if (kLds) {
    stage_scene(sc, lds, threadIdx.x, kBlock);
    __syncthreads();
  }
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  V3 o = mk(0, 0, 0), d = mk(1, 1, 1);
  if (valid) {
    o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
  }
  V3 dn = unit(d);
  HitRec h = trace_any<kLds>(sc, lds, valid, o, d, dn, wsh[threadIdx.x >> 6]);
  if (!valid) return;
  prim[i] = hit_prim(sc, h);
  t[i] = h.t;
  V3 p = mk(0, 0, 0), nn = mk(0, 0, 0);
  if (h.leaf >= 0) hit_point(sc, h, o, dn, p, nn);
  if (pos) { pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z; }
  if (nrm) { nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z; }
}

// The same for scenes that do not fit LDS: a fixed number of resident waves pull rays
// from the arrays through one counter (trace_stream), so no lane waits for the
// longest ray of its wave.  *next must be 0 at launch.
RTR_DEV uint32_t wave_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct RayArrays {
  typedef unsigned long long Index;
  typedef unsigned long long Tag;
  const DeviceScene &sc;
  const float *org, *dir;
  unsigned long long n;
  int32_t *prim;
  float *t, *pos, *nrm;
  unsigned long long *next;
  RTR_DEV unsigned grab(Index &base, bool &more) const {
    unsigned long long b = 0;
    if ((threadIdx.x & 63u) == 0) b = atomicAdd(next, 64ull);
    b = ((unsigned long long)wave_first((uint32_t)(b >> 32)) << 32) | wave_first((uint32_t)b);
    base = b;
    if (b + 64 >= n) more = false;
    if (b >= n) return 0;
    const unsigned long long left = n - b;
    return left < 64 ? (unsigned)left : 64u;
  }
  RTR_DEV void report(unsigned long long, unsigned long long) const {}
  RTR_DEV void load(Index i, V3 &o, V3 &d, Tag &tag) const {
    o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    tag = i;
  }
  RTR_DEV void store(Tag i, int leaf, float th, V3 p, V3 nn) const {
    HitRec h{th, 0.0f, 0.0f, leaf};
    prim[i] = hit_prim(sc, h);
    t[i] = th;
    if (pos) { pos[3 * i] = p.x; pos[3 * i + 1] = p.y; pos[3 * i + 2] = p.z; }
    if (nrm) { nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z; }
  }
};

#ifdef RTR_STREAM_WAVES
#define RTR_STREAM_ATTR __attribute__((amdgpu_waves_per_eu(RTR_STREAM_WAVES, RTR_STREAM_WAVES)))
#else
#define RTR_STREAM_ATTR
#endif
__global__ void __launch_bounds__(kBlock) RTR_STREAM_ATTR
k_trace_api_stream(DeviceScene sc, const float *__restrict__ org, const float *__restrict__ dir,
                   unsigned long long n, int32_t *__restrict__ prim, float *__restrict__ t,
                   float *__restrict__ pos, float *__restrict__ nrm, unsigned long long *next) {
  __shared__ StreamShared sh;
  static_assert(kBlock == 256, "StreamShared holds four waves");
  if (threadIdx.x == 0) sh.merge_lock = 0;
  __syncthreads();
  RayArrays src{sc, org, dir, n, prim, t, pos, nrm, next};
  trace_stream(sc, src, sh);
}

// ---- self-test of the exact-arithmetic shortcuts ----
// reciprocal_pair against the IEEE division for every float: workgroup b covers the 2^16
// bit patterns b * 65536 ... (launched with 65536 workgroups of 256 threads).
__global__ void __launch_bounds__(kBlock)
k_selftest_reciprocal(unsigned long long *__restrict__ mismatches) {
  unsigned bad = 0;
  for (unsigned k = 0; k < 256u; k += 2u) {
    const uint32_t u = blockIdx.x * 65536u + k * 256u + threadIdx.x;
    const f32x2 d = {__uint_as_float(u), __uint_as_float(u + 256u)};
    const f32x2 got = reciprocal_pair(d);
    const float w0 = 1.0f / d.x, w1 = 1.0f / d.y;
    // any NaN equals any NaN
    bad += (__float_as_uint(got.x) != __float_as_uint(w0) && !(got.x != got.x && w0 != w0)) ? 1u : 0u;
    bad += (__float_as_uint(got.y) != __float_as_uint(w1) && !(got.y != got.y && w1 != w1)) ? 1u : 0u;
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// div_by_pi against the IEEE quotient x / (float)pi for all 2^32 floats (both signs of zero, the
// numerators outside its range, infinities and NaNs included): thread t of workgroup b takes the
// bit patterns b * 65536 + 3 k * 256 + t, three at a time as one colour, so that the lanes of a
// wave hold neighbouring patterns and the wave's guard is met wherever the range is.
__global__ void __launch_bounds__(kBlock)
k_selftest_div_by_pi(unsigned long long *__restrict__ mismatches) {
  unsigned bad = 0;
  const float pi = (float)kPi;
  for (unsigned k = 0; k + 3u <= 258u; k += 3u) {  // (the last colour runs two patterns into the next workgroup's)
    const uint32_t u = blockIdx.x * 65536u + k * 256u + threadIdx.x;
    const V3 x = mk(__uint_as_float(u), __uint_as_float(u + 256u), __uint_as_float(u + 512u));
    const V3 got = div_by_pi(x);
    const float w[3] = {x.x / pi, x.y / pi, x.z / pi}, g[3] = {got.x, got.y, got.z};
    for (int c = 0; c < 3; c++)  // any NaN equals any NaN
      bad += (__float_as_uint(g[c]) != __float_as_uint(w[c]) && !(g[c] != g[c] && w[c] != w[c])) ? 1u : 0u;
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// sqrt_ranged against sqrtf for every float in [2^-100, 2^100]: thread i of the launch takes
// bit pattern lo + i (launched over the 200 * 2^23 patterns of that range).
__global__ void __launch_bounds__(kBlock)
k_selftest_sqrt(uint32_t lo, uint32_t count, unsigned long long *__restrict__ mismatches) {
  const uint32_t i = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  if (i >= count) return;
  const float s = __uint_as_float(lo + i);
  if (__float_as_uint(sqrt_ranged(s)) != __float_as_uint(sqrtf(s))) atomicAdd(mismatches, 1ull);
  if (i == 0) {
    const float ones = 1.0f / sqrtf(3.0f);  // (the compiler folds neither: IEEE operations on the device)
    if (__float_as_uint(ones) != __float_as_uint(kUnitOfOnes)) atomicAdd(mismatches, 1ull);
    // near_light's constant: the first float whose root, as a double, is above 1e-3
    const float t = near_light_threshold(), below = __uint_as_float(__float_as_uint(t) - 1u);
    if (!((double)sqrtf(t) > 1e-3 && (double)sqrtf(below) <= 1e-3)) atomicAdd(mismatches, 1ull);
  }
}
// unit() against its IEEE form on vectors built to be hard: components next to the values
// whose quotient by the length falls on a midpoint between two floats, plain random
// components, zeros, and vectors outside the fast path's range (tiny and huge lengths,
// components below 2^-60), for which the wave must fall back.
RTR_DEV uint32_t selftest_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void __launch_bounds__(kBlock)
k_selftest_unit(uint32_t round, unsigned long long *__restrict__ mismatches) {
  const uint32_t id = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const uint32_t r0 = selftest_mix(id * 2654435761u + round), r1 = selftest_mix(r0 + 0x9e3779b9u),
                 r2 = selftest_mix(r1 ^ round), r3 = selftest_mix(r2 + id);
  // whole workgroups share a kind, so that some waves take the fast path and some do not
  const uint32_t kind = blockIdx.x & 7u;
  const int e = (kind == 6) ? -70 : (kind == 7) ? 70 : (int)(r0 % 41u) - 20;
  const float scale = __uint_as_float((uint32_t)(127 + e) << 23);
  V3 a = mk(__uint_as_float(0x3f000000u | (r1 >> 9)) * scale, __uint_as_float(0x3f000000u | (r2 >> 9)) * scale,
            __uint_as_float(0x3f000000u | (r3 >> 9)) * scale);
  if (r0 & 0x100u) a.x = -a.x;
  if (r0 & 0x200u) a.y = -a.y;
  if (kind == 1) a.z = (r0 & 0x400u) ? 0.0f : -0.0f;
  if (kind == 2) { a.y = 0.0f; a.z = -0.0f; }
  if (kind == 3) a.z = __uint_as_float(r3 & 0x807fffffu);  // a denormal or zero component
  if (kind == 4 || kind == 5) {
    // x next to q * l for q a midpoint between two floats below one
    const float l0 = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    const uint32_t qb = 0x3e800000u | (r3 >> 10);
    const double q = ((double)__uint_as_float(qb) + (double)__uint_as_float(qb + 1u)) * 0.5;
    a.x = __uint_as_float(__float_as_uint((float)(q * (double)l0)) + (kind == 4 ? 0u : 1u));
  }
  const V3 got = unit(a);
  const float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
  const V3 want = mk(a.x / l, a.y / l, a.z / l);
  auto differ = [](float g, float w) { return __float_as_uint(g) != __float_as_uint(w) && !(g != g && w != w); };
  if (differ(got.x, want.x) || differ(got.y, want.y) || differ(got.z, want.z)) atomicAdd(mismatches, 1ull);
}

// unit_near's two steps against sqrtf and the IEEE quotient, exhaustively: every sum of squares of
// its window (s_lo ... s_hi), and for each the quotient of every float of magnitude zero or in
// [2^-100, 2] by the length.  Thread i takes the numerators i, i + stride, ...
__global__ void __launch_bounds__(kBlock)
k_selftest_unit_near(uint32_t s_lo, uint32_t s_hi, unsigned long long *__restrict__ mismatches) {
  const uint32_t tid = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  unsigned bad = 0;
  for (uint32_t sb = s_lo; sb <= s_hi; sb++) {
    const float s = __uint_as_float(sb);
    const float l = near_one_sqrt(s);
    if (__float_as_uint(l) != __float_as_uint(sqrtf(s))) bad++;
    const NearOne d(l);
    // 2^32 / (grid size) numerators per thread; the launch has 2^20 threads
    for (uint32_t k = 0; k < 4096u; k++) {
      const uint32_t au = k * (1u << 20) + tid;
      const uint32_t mag = au & 0x7fffffffu;
      if (mag != 0u && (mag < 0x0d800000u || mag > 0x40000000u)) continue;  // tiny and not zero, or above 2
      const float a = __uint_as_float(au);
      if (__float_as_uint(d.quotient(a)) != __float_as_uint(a / l)) bad++;
    }
  }
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// div_div against the six IEEE divisions it stands for: colours with components of either sign
// and zeros of either sign over the short way's whole range of exponents, divisors over theirs;
// and, by workgroup, what must send the wave to the IEEE divisions instead -- a tiny component, a
// huge or infinite one, a divisor outside [2^-20, 2^20].  (div_corrected by itself was compared
// with the IEEE quotient for every pair of significands: tools/ubench/exact_division.hip.)
__global__ void __launch_bounds__(kBlock)
k_selftest_div_div(uint32_t round, unsigned long long *__restrict__ mismatches) {
  const uint32_t id = blockIdx.x * (uint32_t)kBlock + threadIdx.x;
  const uint32_t r0 = selftest_mix(id * 2654435761u + round), r1 = selftest_mix(r0 + 0x9e3779b9u),
                 r2 = selftest_mix(r1 ^ round), r3 = selftest_mix(r2 + id), r4 = selftest_mix(r3 ^ 0x85ebca6bu),
                 r5 = selftest_mix(r4 + round);
  const uint32_t kind = blockIdx.x & 7u;
  auto number = [](uint32_t r, int e_lo, int e_hi) {  // random sign and significand, exponent in [e_lo, e_hi]
    const int e = e_lo + (int)((r >> 24) % (uint32_t)(e_hi - e_lo + 1));
    return __uint_as_float((r & 0x807fffffu) | ((uint32_t)(127 + e) << 23));
  };
  V3 x = mk(number(r0, -60, 59), number(r1, -60, 59), number(r2, -60, 59));
  float d1 = fabsf(number(r3, -20, 19)), d2 = fabsf(number(r4, -20, 19));
  if (kind == 1) x.y = (r5 & 1u) ? 0.0f : -0.0f;
  if (kind == 2) { x.x = -0.0f; x.y = 0.0f; x.z = (r5 & 2u) ? 0.0f : -0.0f; }
  if (kind == 3) { d1 = (r5 & 1u) ? 9.5367431640625e-07f : 1048576.0f; d2 = (r5 & 2u) ? 9.5367431640625e-07f : 1048576.0f; }
  if (kind == 4) x.z = __uint_as_float(r5 & 0x807fffffu);                   // a denormal or zero component
  if (kind == 5) x.x = (r5 & 1u) ? __int_as_float(0x7f800000) : 3.0e30f;    // infinite or huge
  if (kind == 6) d1 = fabsf(number(r3, (r5 & 1u) ? -60 : 21, (r5 & 1u) ? -21 : 60));
  if (kind == 7) x = mk(number(r0, -60, -50), number(r1, 50, 59), number(r2, -60, 59));  // the corners of the range
  const V3 got = div_div(x, d1, d2);
  const V3 want = mk(x.x / d1 / d2, x.y / d1 / d2, x.z / d1 / d2);
  auto differ = [](float g, float w) { return __float_as_uint(g) != __float_as_uint(w) && !(g != g && w != w); };
  if (differ(got.x, want.x) || differ(got.y, want.y) || differ(got.z, want.z)) atomicAdd(mismatches, 1ull);
}

// ---- the algorithmic work of a ray batch ----
// What the reference's traversal does for each ray, counted: boxes tested (every node the
// recursion of src/BVH.cpp:66-80 enters), and triangles / spheres tested (leaves whose box
// the ray's line crosses).  One ray per lane, plain walk; a measurement aid (the traversal
// kernels above are priced against these counts), not a fast path.
__global__ void __launch_bounds__(kBlock)
k_count_work(DeviceScene sc, const float *__restrict__ org, const float *__restrict__ dir, size_t n,
             unsigned long long *__restrict__ out3) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  unsigned long long boxes = 0, tris = 0, sphs = 0;
  if (i < n) {
    const V3 o = mk(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    const V3 dn = unit(mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    const GlobalGeom g = global_geom(sc);
    for (int k = 0; k < sc.n_nodes;) {
      const float4 lo = g.node_lo(k), hi = g.node_hi(k);
      const int leaf = (int)wbits(hi);
      boxes++;
      if (!slab(lo, hi, o, dn)) {
        k = (int)wbits(lo);
      } else if (leaf < 0) {
        k++;
      } else {
        if (leaf >> 30) sphs++; else tris++;
        k = (int)wbits(lo);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    boxes += __shfl_down(boxes, off);
    tris += __shfl_down(tris, off);
    sphs += __shfl_down(sphs, off);
  }
  if (lane_id() == 0) {
    atomicAdd(&out3[0], boxes);
    atomicAdd(&out3[1], tris);
    atomicAdd(&out3[2], sphs);
  }
}

// ---- Scene::checkConnectable, include/Scene.h:49-56 ----
// Closest hit from a towards b; connected iff something is hit and its distance is
// within 1e-3 of |b - a| (the reference's ESP argument is not used by its body).
template <bool kLds>
__global__ void __launch_bounds__(kBlock)
k_check_connectable(DeviceScene sc, const float *__restrict__ a3, const float *__restrict__ b3, size_t n,
                    unsigned char *__restrict__ connected) {
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  if (kLds) {
    stage_scene(sc, lds, threadIdx.x, kBlock);
    __syncthreads();
  }
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  V3 a = mk(0, 0, 0), ab = mk(1, 1, 1);
  if (valid) {
    a = mk(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
    ab = mk(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]) - a;
  }
  const V3 d = unit(ab);  // Ray(a, (b - a).normalize())
  HitRec h = trace_any<kLds>(sc, lds, valid, a, d, unit(d), wsh[threadIdx.x >> 6]);
  if (!valid) return;
  // std::abs(float) against the double literal 1e-3
  connected[i] = (h.leaf >= 0 && (double)fabsf(h.t - length(ab)) < 1e-3) ? 1 : 0;
}

// The same for scenes that do not fit LDS (trace_stream); *next must be 0 at launch.
struct ConnectPairs {
  typedef unsigned long long Index;
  typedef unsigned long long Tag;
  const float *a3, *b3;
  unsigned long long n;
  unsigned char *connected;
  unsigned long long *next;
  RTR_DEV unsigned grab(Index &base, bool &more) const {
    unsigned long long b = 0;
    if ((threadIdx.x & 63u) == 0) b = atomicAdd(next, 64ull);
    b = ((unsigned long long)wave_first((uint32_t)(b >> 32)) << 32) | wave_first((uint32_t)b);
    base = b;
    if (b + 64 >= n) more = false;
    if (b >= n) return 0;
    const unsigned long long left = n - b;
    return left < 64 ? (unsigned)left : 64u;
  }
  RTR_DEV void report(unsigned long long, unsigned long long) const {}
  RTR_DEV V3 span(Index i, V3 &a) const {
    a = mk(a3[3 * i], a3[3 * i + 1], a3[3 * i + 2]);
    return mk(b3[3 * i], b3[3 * i + 1], b3[3 * i + 2]) - a;
  }
  RTR_DEV void load(Index i, V3 &o, V3 &d, Tag &tag) const {
    d = unit(span(i, o));  // Ray(a, (b - a).normalize())
    tag = i;
  }
  RTR_DEV void store(Tag i, int leaf, float t, V3, V3) const {
    V3 a;
    const V3 ab = span(i, a);
    connected[i] = (leaf >= 0 && (double)fabsf(t - length(ab)) < 1e-3) ? 1 : 0;
  }
};

__global__ void __launch_bounds__(kBlock) RTR_STREAM_ATTR
k_check_connectable_stream(DeviceScene sc, const float *__restrict__ a3, const float *__restrict__ b3,
                           unsigned long long n, unsigned char *__restrict__ connected,
                           unsigned long long *next) {
  __shared__ StreamShared sh;
  if (threadIdx.x == 0) sh.merge_lock = 0;
  __syncthreads();
  ConnectPairs src{a3, b3, n, connected, next};
  trace_stream(sc, src, sh);
}

// ---- traversal of the primary rays (one per pixel, shared by all its samples) ----
template <bool kLds>
__global__ void __launch_bounds__(kBlock)
k_trace_primary(DeviceScene sc, CameraSetup cam, const float4 *__restrict__ dir, int n,
                float4 *__restrict__ hit) {
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  if (kLds) {
    stage_scene(sc, lds, threadIdx.x, kBlock);
    __syncthreads();
  }
  int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  V3 d = valid ? xyz(dir[i]) : mk(1, 1, 1);
  HitRec h = trace_any<kLds>(sc, lds, valid, cam.eye, d, unit(d), wsh[threadIdx.x >> 6]);
  if (valid) hit[i] = make_float4(h.t, h.b1, h.b2, __int_as_float(h.leaf));
}

// ---- traversal of the ray queue ----
template <bool kLds>
__global__ void __launch_bounds__(kBlock)
k_trace_queue(DeviceScene sc, PathPool p, const uint32_t *__restrict__ queue,
              BlockCounters *__restrict__ bc, int parity, unsigned grp_off) {
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  // workgroup -> (group of kGroup queue segments, part of their concatenation).
  // Workgroups are dealt to the 8 XCDs round-robin.  Numbering (group, part) with
  // the group fastest puts equally many full, partial and empty parts on every
  // XCD; with the part fastest, all the empty last parts landed on one XCD and the
  // partly filled ones on another, idling a quarter of the chip.
  const unsigned n_grp = gridDim.x / kGroupParts;
  const unsigned grp = grp_off + blockIdx.x % n_grp;
  const unsigned gpart = blockIdx.x / n_grp;
  // shadow rays and extension rays are separate streams: a wave then holds one
  // kind only (shadow rays all aim at the lights and share most of their walk)
  const unsigned kind = gpart / kStreamParts;
  const unsigned part = gpart % kStreamParts;
  unsigned cnt[kGroup], nq = 0;
#pragma unroll
  for (int g = 0; g < kGroup; g++) {
    cnt[g] = bc[grp * kGroup + g].qcount[parity][kind];
    nq += cnt[g];
  }
  if (part * kBlock >= nq) return;  // whole workgroup has nothing to do
  if (part == 0 && threadIdx.x < kGroup)
    bc[grp * kGroup + threadIdx.x].rays[kind] += bc[grp * kGroup + threadIdx.x].qcount[parity][kind];  // only writer
  if (kLds) {
    stage_scene(sc, lds, threadIdx.x, kBlock);
    __syncthreads();
  }
  unsigned r = part * kBlock + threadIdx.x;  // index in the packed stream of the group
  const bool valid = r < nq;
  uint32_t slot = 0;
  bool ext = false;
  V3 o = mk(0, 0, 0), d = mk(1, 1, 1);
  if (valid) {
    unsigned seg = grp * kGroup;
#pragma unroll
    for (int g = 0; g < kGroup - 1; g++) {
      if (r >= cnt[g]) { r -= cnt[g]; seg++; } else break;
    }
    uint32_t e = queue[(size_t)seg * kSegment + kind * kLogicBlock + r];
    slot = e & 0x7fffffffu;
    ext = e >> 31;
    o = xyz(p.A[slot]);
    d = xyz(ext ? p.K[slot] : p.J[slot]);
  }
  const V3 dn = unit(d);
  HitRec h = trace_any<kLds>(sc, lds, valid, o, d, dn, wsh[threadIdx.x >> 6]);
  if (!valid) return;
  // hand the path logic world-space results, so it needs neither the primitive
  // nor the ray again: Intersection::postion / ::normal of src/Trangle.cpp:23-24,
  // src/Sphere.cpp:25-26
  V3 hp = mk(__int_as_float(0x7fc00000), 0, 0), hn = mk(0, 0, 0);
  if (h.leaf >= 0) hit_point(sc, h, o, dn, hp, hn);
  if (ext) {
    p.he[slot] = pack(hp, (uint32_t)hit_prim(sc, h));
    if (h.leaf >= 0) p.he2[slot] = pack(unit(hn), 0.0f);
  } else {
    p.hs[slot] = pack(hp, h.t);
  }
}

// The same for scenes that do not fit LDS (trace_stream): resident waves pull the queue
// segments of the launch, one (k_logic workgroup, ray kind) at a time, through
// cnt->trace_next[half], and refill their lanes from them.
struct QueueRays {
  typedef uint32_t Index;  // position in the queue array
  typedef uint32_t Tag;    // queue entry: pool slot | extension ray << 31
  const DeviceScene &sc;
  PathPool p;
  const uint32_t *queue;
  BlockCounters *bc;
  unsigned int *next;
  int parity;
  unsigned blk_begin, n_pairs;  // first k_logic workgroup of the launch; 2 x their number
  // A wave reserves `take` consecutive (workgroup, ray kind) segments with one atomic add and
  // works through the non-empty ones.  take = 1 while segments are full (finest balance: rays
  // differ in cost by orders of magnitude); when most are empty -- the last iterations of a
  // render, when few paths are left -- a single counter handed out one segment at a time is
  // what a launch waits for (16 K atomic adds in a row on one address: 0.25 ms for no rays),
  // so the reservation grows to up to 64.
  unsigned take;
  unsigned pos, end;            // wave-uniform: unread entries of the current segment
  unsigned long long pending;   // reserved segments with rays, not started yet (bit = lane)
  unsigned my_pos, my_cnt;      // lane l: segment l of the reservation
  Counters *totals;
  RTR_DEV void report(unsigned long long boxes, unsigned long long prims) const {
    if (lane_id() == 0 && boxes) {
      atomicAdd(&totals->boxes, boxes);
      atomicAdd(&totals->prims, prims);
    }
  }
  RTR_DEV unsigned grab(Index &base, bool &more) {
    while (pos == end) {
      if (pending == 0) {
        unsigned w = 0;
        if (lane_id() == 0) w = atomicAdd(next, take);
        w = wave_first(w);
        if (w >= n_pairs) {
          more = false;
          return 0;
        }
        const unsigned idx = w + lane_id();
        const unsigned b = blk_begin + (idx >> 1), kind = idx & 1u;
        unsigned c = 0;
        if (lane_id() < take && idx < n_pairs) c = bc[b].qcount[parity][kind];
        if (c) bc[b].rays[kind] += c;  // only writer
        my_pos = b * (unsigned)kSegment + kind * (unsigned)kLogicBlock;
        my_cnt = c;
        pending = __ballot(c != 0);
        continue;
      }
      const int l = __ffsll((long long)pending) - 1;
      pending &= pending - 1;
      pos = (unsigned)__builtin_amdgcn_readlane((int)my_pos, l);
      end = pos + (unsigned)__builtin_amdgcn_readlane((int)my_cnt, l);
    }
    base = pos;
    const unsigned got = end - pos < 64u ? end - pos : 64u;
    pos += got;
    return got;
  }
  RTR_DEV void load(Index i, V3 &o, V3 &d, Tag &tag) const {
    const uint32_t e = queue[i];
    const uint32_t slot = e & 0x7fffffffu;
    tag = e;
    o = xyz(p.A[slot]);
    d = xyz((e >> 31) ? p.K[slot] : p.J[slot]);
  }
  // world-space results, so the path logic needs neither the primitive nor the ray again
  RTR_DEV void store(Tag e, int leaf, float t, V3 hp, V3 hn) const {
    const uint32_t slot = e & 0x7fffffffu;
    if (leaf < 0) hp = mk(__int_as_float(0x7fc00000), 0, 0);
    if (e >> 31) {
      HitRec h{t, 0.0f, 0.0f, leaf};
      p.he[slot] = pack(hp, (uint32_t)hit_prim(sc, h));
      if (leaf >= 0) p.he2[slot] = pack(unit(hn), 0.0f);
    } else {
      p.hs[slot] = pack(hp, t);
    }
  }
};

__global__ void __launch_bounds__(kBlock) RTR_STREAM_ATTR
k_trace_queue_stream(DeviceScene sc, PathPool p, const uint32_t *__restrict__ queue,
                     BlockCounters *__restrict__ bc, Counters *__restrict__ cnt, int parity,
                     unsigned blk_begin, unsigned n_blk) {
  __shared__ StreamShared sh;
  if (threadIdx.x == 0) sh.merge_lock = 0;
  __syncthreads();
  // average filling of the launch's segments (each holds up to kLogicBlock rays)
  const unsigned n_pairs = 2 * n_blk;
  const unsigned queued = wave_first(cnt->queued[blk_begin != 0][parity]);
  const unsigned per_pair = queued / n_pairs;
  const unsigned take = per_pair >= 64u ? 1u : per_pair >= 16u ? 4u : per_pair >= 4u ? 16u : 64u;
  QueueRays src{sc, p, queue, bc, &cnt->trace_next[blk_begin != 0], parity, blk_begin, n_pairs, take, 0, 0, 0ull, 0, 0, cnt};
  trace_stream(sc, src, sh);
}

// ---- first vertex of every pixel; pixels that need no sampling are finished here ----
__global__ void __launch_bounds__(kBlock)
k_classify(RenderArgs a, DeviceScene sc, const DeviceMaterial *__restrict__ mats, CameraSetup cam,
           const float4 *__restrict__ dir, const float4 *__restrict__ hit, PixelCache pc,
           uint32_t *__restrict__ shade_list, Counters *__restrict__ cnt,
           float4 *__restrict__ partial) {
  int lp = blockIdx.x * kBlock + threadIdx.x;
  bool shade = false;
  if (lp < a.n_pix) {
    float4 dv = dir[lp];
    float4 hv = hit[lp];
    HitRec h{hv.x, hv.y, hv.z, __float_as_int(hv.w)};
    V3 d = xyz(dv);
    V3 dn = unit(d);
    V3 value = mk(0, 0, 0);
    int mat = -1;
    if (h.leaf >= 0) {
      V3 pos, nn;
      hit_point(sc, h, cam.eye, dn, pos, nn);
      int m = sc.prim_material[hit_prim(sc, h)];
      DeviceMaterial dm = mats[m];
      if (mat_emissive(dm)) {
        mat = -2;
        value = xyz(dm.e_kind);
      } else {
        mat = m;
        shade = true;
        pc.nrm[lp] = pack(unit(nn), 0.0f);
      }
      pc.pos_mat[lp] = pack(pos, (uint32_t)mat);
    } else {
      pc.pos_mat[lp] = pack(mk(0, 0, 0), (uint32_t)mat);
    }
    // Material's unit(wi) of wi = -dir.normalize() (src/Render.cpp:82): the same for every
    // sample of the pixel, so formed here once (shade_vertex<true> takes it as it is)
    pc.wi[lp] = pack(unit(-dn), wbits(dv));
    if (!shade) {
      // every sample of this pixel returns `value` (src/Render.cpp:84-87); the
      // sums are still formed by repeated addition like the reference's loop
      for (int c = 0; c < a.n_chunks; c++) {
        int k0 = c * a.chunk_spp, k1 = min(a.spp, k0 + a.chunk_spp);
        V3 acc = mk(0, 0, 0);
        for (int k = k0; k < k1; k++) acc = acc + value;
        partial[(size_t)c * a.n_pix + lp] = pack(acc, 0.0f);
      }
    }
  }
  unsigned long long m = __ballot(shade);
  if (m) {
    unsigned base = 0;
    int leader = __ffsll((long long)m) - 1;
    if ((int)lane_id() == leader) base = atomicAdd(&cnt->n_shade, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (shade) shade_list[base + __popcll(m & lanes_below())] = (uint32_t)lp;
  }
}

// ---- shading of one path vertex: src/Render.cpp:90-116 ----
// Next-event estimation (light sample, shadow ray) and, if Russian roulette lets
// the path go on, the BSDF sample (extension ray).  Random numbers are drawn in
// the reference's order.
struct Emit {
  bool has_shadow, has_ext;
  V3 l;          // shadow ray direction, unit(light point - pos)
  V3 ln, wn;     // unit(l), unit(w1): the directions traversal normalises again
  V3 light_pos;  // the sampled light point
  V3 weight;     // Le * f * cos * cos'
  float light_pdf;
  V3 w1;         // extension ray direction
  V3 beta_next;  // path weight after the bounce (zero if the path stops here)
  V3 uo_ext;     // unit(wn) as Material::eval formed it for the extension direction
};

// kUnitWi: `wi` is unit(wi) already.  The reference's wi at a vertex reached by an extension ray
// is the negated ray direction wn (src/Render.cpp:82), and Material::eval normalised that very
// direction when the ray was made: unit(-wn) = -unit(wn) bit for bit (the length is the same
// and IEEE division is odd in its numerator), so a caller that keeps e.uo_ext need not
// normalise again; the first vertex of a sample has its unit(wi) formed once per work unit.
template <bool kUnitWi = false, class Tab>
RTR_DEV void shade_vertex(const RenderArgs &a, const Tab &T, V3 pos, V3 N, V3 wi, V3 beta,
                          uint32_t mat, uint32_t gpix, uint32_t k, uint32_t depth, Emit &e) {
  DeviceMaterial dm = T.mat(mat);
  // (N is the normalised normal of the hit: its second normalisation takes the short way)
  const V3 ui = kUnitWi ? wi : unit(wi), un = unit_near(N);  // shared by the two evaluations below
  VertexStream rng(a.seed, gpix, k, depth);
  rng.round_keys = T.philox_keys();
  LightSample ls;
  sample_light(T, rng, ls);
  e.has_shadow = false;
  e.has_ext = false;
  e.light_pos = ls.pos;
  e.light_pdf = ls.pdf;
  if (ls.pdf > 0) {
    V3 l = unit(ls.pos - pos);
    V3 l2 = unit_near(l);
    V3 Le = xyz(T.light_le(ls.light));
    e.weight = Le * mat_eval_units(dm, ui, l2, un) * dot(N, l2) * dot(unit_near(ls.nrm), -l2);
    e.l = l;
    e.ln = l2;
    e.has_shadow = true;
  }
  if (rng.next() < a.rr) {
    V3 wo;
    if (mat_kind(dm) == 1) {
      wo = mat_mirror_units(ui, un);
    } else {
      float u1 = rng.next();
      float u2 = rng.next();
      wo = mat_sample_uv(N, u1, u2);
    }
    float pdf = mat_pdf(dm);
    V3 w1 = unit_near(wo);  // a sampled or mirrored direction: of unit length up to rounding
    e.wn = unit_near(w1);
    e.beta_next = div_div(beta * mat_eval_units(dm, ui, e.wn, un, e.uo_ext) * dot(N, w1), pdf, a.rr);
    e.w1 = w1;
    e.has_ext = true;
  } else {
    e.beta_next = mk(0, 0, 0);
  }
}

// ---- one wavefront iteration of path logic ----
__global__ void __launch_bounds__(kLogicBlock)
k_logic(RenderArgs a, DeviceScene sc, DeviceLights L, const DeviceMaterial *__restrict__ mats,
        PixelCache pc, PathPool p, const uint32_t *__restrict__ shade_list,
        uint32_t *__restrict__ queue, BlockCounters *__restrict__ bc, float4 *__restrict__ partial,
        Counters *__restrict__ cnt, uint2 *__restrict__ wave_res, int parity, int block_off) {
  __shared__ unsigned s_count[4][kLogicWaves];
  const int blk = (int)blockIdx.x + block_off;  // the pool may be driven as several launches
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    cnt->trace_next[block_off != 0] = 0;
    cnt->queued[block_off != 0][parity ^ 1] = 0;  // its reader, the previous traversal launch, is over
  }
  const int slot = blk * kLogicBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const bool in_range = slot < a.pool;

  uint32_t flags = 0;
  V3 pos = mk(0, 0, 0), N = pos, wi = pos, beta = pos, Lp = pos, acc = pos;
  uint32_t mat = 0, depth = 0, lp = 0, k = 0, k_end = 0, unit_id = 0xffffffffu, gpix = 0;
  if (in_range && !a.first) {
    float4 h4 = p.Hh[slot];
    float4 f4 = p.F[slot];
    flags = wbits(h4);
    acc = xyz(f4); unit_id = wbits(f4);
    if (flags & kHasUnit) {
      float4 a4 = p.A[slot], b4 = p.B[slot], c4 = p.C[slot], d4 = p.D[slot], e4 = p.E[slot];
      pos = xyz(a4); mat = wbits(a4);
      N = xyz(b4); depth = wbits(b4);
      wi = xyz(c4); lp = wbits(c4);
      beta = xyz(d4); k = wbits(d4);
      Lp = xyz(e4); k_end = wbits(e4);
    }
    if (flags & kLive) {
      // everything the two results need is fetched in one go: one memory round
      // trip instead of one per dependent branch
      const float4 hs4 = p.hs[slot], he4 = p.he[slot], he24 = p.he2[slot], g4 = p.G[slot],
                   k4 = p.K[slot], i4 = p.I[slot];
      gpix = wbits(i4);
      // next-event result: src/Render.cpp:95-107.  A miss arrives as a NaN point,
      // which fails the distance test like the reference's `testInter.hit` does.
      if (flags & kHasShadow) {
        if (near_light(xyz(hs4) - xyz(h4))) {
          V3 term = div_div(xyz(g4), hs4.w * hs4.w, g4.w);
          Lp = Lp + beta * term;
        }
      }
      // extension result: src/Render.cpp:117-125
      bool cont = false;
      if (flags & kHasExt) {
        int prim2 = (int)wbits(he4);
        if (prim2 >= 0) {
          int m2 = sc.prim_material[prim2];
          if (!mat_emissive(mats[m2]) && !(a.max_depth > 0 && (int)depth + 1 >= a.max_depth)) {
            beta = xyz(i4);
            wi = -unit(unit(xyz(k4)));  // unit(wi) of wi = -unit(w1)
            N = xyz(he24);
            pos = xyz(he4);
            mat = (uint32_t)m2;
            depth++;
            cont = true;
          }
        }
      }
      if (!cont) {
        acc = acc + Lp;  // color = color + pathTracing(...), src/Render.cpp:659
        k++;
        flags &= ~kLive;
      }
    }
  }

  // A slot without a live path starts the next sample of its unit, or asks for a new
  // unit.  Units are handed out on demand, as in the fused kernel: a wave reserves
  // 64 consecutive units with one atomic add on the global cursor, keeps what it has
  // not dealt yet in its record of `wave_res` between launches, and deals by ballot
  // rank.  (A static split per slot let the slots with the longest paths run on for
  // a third more iterations than the average, with the pool mostly empty.)
  // Unit u is chunk u / n_shade of pixel-list entry u % n_shade; 0xffffffff: none, and
  // none to come.
  bool unit_finished = in_range && (flags & kHasUnit) && !(flags & kLive) && k >= k_end;
  bool unit_changed = false;
  if (unit_finished) {
    unsigned chunk = unit_id / a.n_shade;
    partial[(size_t)chunk * a.n_pix + lp] = pack(acc, 0.0f);
    flags &= ~kHasUnit;
  }
  {
    const unsigned gw = (unsigned)slot >> 6;  // this wave's record
    uint32_t res_base = 0, res_left = 0;
    if (!a.first) {
      const uint2 r = wave_res[gw];
      res_base = r.x;
      res_left = r.y;
    }
    bool drained = !a.first && res_left == 0 && res_base == 0xffffffffu;
    bool touched = false;
    bool need = in_range && !(flags & kHasUnit) && (a.first || unit_id != 0xffffffffu);
    for (unsigned long long want = __ballot(need); want != 0; want = __ballot(need)) {
      touched = true;
      if (res_left == 0) {
        if (!drained) {
          unsigned long long base = 0;
          if (lane_id() == 0) base = atomicAdd(&cnt->next_unit, 64ull);
          base = __shfl(base, 0);
          if (base >= a.n_units) {
            drained = true;
          } else {
            res_base = (uint32_t)base;
            res_left = min(64u, a.n_units - res_base);
          }
        }
        if (drained) {
          // nothing left anywhere: these slots are done for good
          if (need) {
            unit_id = 0xffffffffu;
            unit_changed = true;
            need = false;
          }
          res_base = 0xffffffffu;
          break;
        }
      }
      const uint32_t rank = (uint32_t)__popcll(want & lanes_below());
      if (need && rank < res_left) {
        unit_id = res_base + rank;
        const uint32_t entry = unit_id % a.n_shade, chunk = unit_id / a.n_shade;
        lp = shade_list[entry];
        k = chunk * (uint32_t)a.chunk_spp;
        k_end = min((uint32_t)a.spp, k + (uint32_t)a.chunk_spp);
        acc = mk(0, 0, 0);
        gpix = wbits(pc.wi[lp]);
        flags |= kHasUnit;
        unit_changed = true;
        need = false;
      }
      const uint32_t dealt = min((uint32_t)__popcll(want), res_left);
      res_base += dealt;
      res_left -= dealt;
    }
    if ((touched || a.first) && lane_id() == 0) wave_res[gw] = make_uint2(res_base, res_left);
  }
  bool live = in_range && (flags & kHasUnit);
  if (live && !(flags & kLive)) {
    // new sample: every sample of a pixel starts at the same first vertex
    float4 pm = pc.pos_mat[lp];
    pos = xyz(pm); mat = wbits(pm);
    N = xyz(pc.nrm[lp]);
    wi = xyz(pc.wi[lp]);
    beta = mk(1, 1, 1);
    Lp = mk(0, 0, 0);
    depth = 0;
    flags |= kLive;
    unit_changed = true;  // acc moved on
  }

  // shade the current vertex: src/Render.cpp:90-116
  bool has_shadow = false, has_ext = false;
  if (live) {
    Emit e;
    shade_vertex<true>(a, GlobalTables{sc, L, mats}, pos, N, wi, beta, mat, gpix, k, depth, e);  // wi: unit(wi)
    has_shadow = e.has_shadow;
    has_ext = e.has_ext;
    if (has_shadow) {
      p.G[slot] = pack(e.weight, e.light_pdf);
      p.J[slot] = pack(e.l, 0.0f);
    }
    p.I[slot] = pack(e.beta_next, gpix);
    if (has_ext) p.K[slot] = pack(e.w1, 0.0f);
    flags = (flags & ~(kHasShadow | kHasExt)) | (has_shadow ? kHasShadow : 0u) |
            (has_ext ? kHasExt : 0u);
    p.A[slot] = pack(pos, mat);
    p.B[slot] = pack(N, depth);
    p.C[slot] = pack(wi, lp);
    p.D[slot] = pack(beta, k);
    p.E[slot] = pack(Lp, k_end);
    if (unit_changed) p.F[slot] = pack(acc, unit_id);
    p.Hh[slot] = pack(e.light_pos, flags);
  } else if (in_range && (unit_changed || a.first)) {
    p.F[slot] = pack(acc, unit_id);
    p.Hh[slot] = pack(mk(0, 0, 0), 0u);
  }

  // Compaction of the emitted rays into this workgroup's queue segment: 64-wide
  // ballots give each lane its rank inside the wave, LDS combines the waves.
  // Shadow rays fill the first half of the segment, extension rays the second.
  unsigned long long ms = __ballot(has_shadow), me = __ballot(has_ext), ml = __ballot(live),
                     mf = __ballot(unit_finished);
  if (lane_id() == 0) {
    s_count[0][wave] = (unsigned)__popcll(ms);
    s_count[1][wave] = (unsigned)__popcll(me);
    s_count[2][wave] = (unsigned)__popcll(ml);
    s_count[3][wave] = (unsigned)__popcll(mf);
  }
  __syncthreads();
  unsigned sh_before = 0, ext_before = 0, sh_total = 0, ext_total = 0;
  for (int w = 0; w < kLogicWaves; w++) {
    unsigned cs = s_count[0][w], ce = s_count[1][w];
    if (w < wave) { sh_before += cs; ext_before += ce; }
    sh_total += cs;
    ext_total += ce;
  }
  if (threadIdx.x == 0) {
    unsigned nl = 0, nf = 0;
    for (int w = 0; w < kLogicWaves; w++) { nl += s_count[2][w]; nf += s_count[3][w]; }
    BlockCounters &b = bc[blk];
    b.qcount[parity][0] = sh_total;
    b.qcount[parity][1] = ext_total;
    if (sh_total + ext_total) atomicAdd(&cnt->queued[block_off != 0][parity], sh_total + ext_total);
    if (a.first) { b.done = nf; b.vertices = nl; b.rays[0] = 0; b.rays[1] = 0; b.qcount[parity ^ 1][0] = 0; b.qcount[parity ^ 1][1] = 0; }
    else { b.done += nf; b.vertices += nl; }
  }
  uint32_t *seg = queue + (size_t)blk * kSegment;
  if (has_shadow) seg[sh_before + __popcll(ms & lanes_below())] = (uint32_t)slot;
  if (has_ext) seg[kLogicBlock + ext_before + __popcll(me & lanes_below())] = (uint32_t)slot | 0x80000000u;
}

// ---- the whole sampling loop in one kernel ----
// For scenes whose traversal needs no per-wave queues (trace_any with staged
// geometry), path state never has to leave the registers: every lane carries one
// path, shades its vertex, traces the shadow ray and the extension ray itself and
// moves on to the next sample or work unit when the path ends.  All lanes of a
// wave do the same thing in every round (shade, trace, trace), whatever the depth
// of their paths, so there is nothing to sort or compact, and no path state,
// ray queue or hit record goes through HBM.
//
// Units are handed out on demand: a wave reserves 64 consecutive units with one
// atomic add on a global cursor and deals them to its lanes as they run out of
// work (ranks from a ballot, no memory traffic), so all lanes of the grid finish
// within one unit of each other -- with a static split the slowest lane of every
// wave, and the slowest wave, held the rest up for several per cent of the run.
// Unit u is (chunk u / n_shade, pixel-list entry u % n_shade).  Which lane takes a
// unit does not matter: a unit's samples are summed in order by whoever owns it, so
// the image is the same as the wavefront loop's, bit for bit.
template <bool kLds>
RTR_DEV void hit_point_any(const DeviceScene &sc, const LdsScene &lds, const HitRec &h, V3 o, V3 dn,
                           V3 &pos, V3 &nrm) {
  const float4 *tris = kLds ? lds.tris : sc.tris;
  const float4 *sphs = kLds ? lds.sphs : sc.sphs;
  int row = h.leaf & 0x3fffffff;
  if (h.leaf >> 30) {
    float4 q = sphs[row];
    pos = o + dn * h.t;
    nrm = unit(pos - xyz(q));
  } else {
    float4 q0 = tris[3 * row], q1 = tris[3 * row + 1], q2 = tris[3 * row + 2];
    pos = xyz(q0) * (1 - h.b1 - h.b2) + xyz(q1) * h.b1 + xyz(q2) * h.b2;
    nrm = mk(q0.w, q1.w, q2.w);
  }
}

// Per-lane state of the current work unit lives in LDS (one column per lane, so
// accesses never conflict): it is touched once per sample, not once per vertex, and
// the registers it would take decide how many waves fit on a SIMD.
struct UnitState {
  float fpos[3][kBlock], fN[3][kBlock], fwi[3][kBlock], acc[3][kBlock];
  uint32_t fmat[kBlock], out[kBlock], k_end[kBlock];  // out: index of the unit's partial sum
};
// Radiance of the lane's current path so far and the lane's statistics: in LDS where the
// workgroup has room for them (the flavour with staged geometry: five registers fewer make its
// fourth wave spill-free), in registers in the tree-walking flavour, whose per-wave rings
// need the space.
template <bool kInLds>
struct UnitStateOf : UnitState {
  float Lp[kInLds ? 3 : 1][kInLds ? kBlock : 1];
  uint32_t n_vertices[kInLds ? kBlock : 1], n_rays[kInLds ? kBlock : 1];
};

// Kernel arguments that are only needed when a lane takes a new work unit or finishes one
// (once in a dozen rounds or more): kept in LDS, so that they do not hold scalar registers
// through the sampling loop -- the loop needs all of them for leaf records and lane masks,
// and what does not fit is spilled into lanes of a vector register.
struct ColdArgs {
  PixelCache pc;
  const uint32_t *shade_list;
  float4 *partial;
  Counters *fc;
  uint32_t n_pix, chunk_spp, spp;
  // the record table of trace_all2: read back (wave-uniform) at the start of every pass
  const float4 *items;
  int32_t n_items, n_pairs;
};

// prim id, material and emissive flag of a hit
template <bool kLds>
RTR_DEV void hit_info(const DeviceScene &sc, const DeviceMaterial *__restrict__ mats,
                      const LdsShade &lsh, const HitRec &h, int node, int &prim, int &mat,
                      bool &emissive) {
  if (kLds) {
    int info = lsh.node_info[node];
    prim = info & 0xffff;
    mat = (info >> 16) & 0x3fff;
    emissive = (info >> 30) & 1;
  } else {
    prim = hit_prim(sc, h);
    mat = sc.prim_material[prim];
    emissive = mat_emissive(mats[mat]);
  }
}

// Four waves per SIMD (128 registers): with three the kernel waits on its own
// instruction latencies (-8 %); the few dozen spilled registers this costs are
// cheaper.  RTR_FUSED_WAVES=3 builds the spill-free variant.
#ifndef RTR_FUSED_WAVES
#define RTR_FUSED_WAVES 4
#endif
// the flavour that walks a tree in memory: four as well (left alone the compiler takes
// 158 registers, three waves: 15 % slower on a 6 K-node scene, 5 % on a 2 M-node one)
#ifndef RTR_FUSED_WAVES_WALK_MIN
#define RTR_FUSED_WAVES_WALK_MIN 4
#define RTR_FUSED_WAVES_WALK_MAX 4
#endif
// kBounded (with kLds): the flavour for scenes whose arithmetic cannot overflow (trace_all2).
template <bool kLds, bool kBounded>
__global__ void __launch_bounds__(kBlock)
    __attribute__((amdgpu_waves_per_eu(kLds ? RTR_FUSED_WAVES : RTR_FUSED_WAVES_WALK_MIN,
                                       kLds ? RTR_FUSED_WAVES : RTR_FUSED_WAVES_WALK_MAX)))
k_render_fused(RenderArgs a, DeviceScene sc, DeviceLights L, const DeviceMaterial *__restrict__ mats,
               const int32_t *__restrict__ node_info, int n_materials, int n_light_tris,
               PixelCache pc, const uint32_t *__restrict__ shade_list, float4 *__restrict__ partial,
               Counters *__restrict__ fc) {
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  __shared__ LdsShade lsh;
  __shared__ UnitStateOf<kLds> us;
  const int tid = threadIdx.x;
  if (kLds) {
    stage_scene(sc, lds, tid, kBlock);
    for (int i = tid; i < n_materials; i += kBlock) lsh.mats[i] = mats[i];
    for (int i = tid; i < L.n_lights; i += kBlock) { lsh.light[i] = L.light[i]; lsh.light_le[i] = L.light_le[i]; }
    for (int i = tid; i < n_light_tris; i += kBlock) lsh.light_tris[i] = L.light_tris[i];
    for (int i = tid; i < sc.n_tris; i += kBlock) lsh.tri_area[i] = L.tri_area[i];
    for (int i = tid; i < sc.n_nodes; i += kBlock) lsh.node_info[i] = node_info[i];
    if (tid < 10) lsh.philox_keys[tid] = make_uint2(a.seed + (uint32_t)tid * 0x9E3779B9u, (uint32_t)tid * 0xBB67AE85u);
    __syncthreads();
  }
  typename WaveLds<kLds>::type &ws = wsh[tid >> 6];
  us.k_end[tid] = 0;
  __shared__ ColdArgs cold;
  if (tid == 0) {
    cold.pc = pc;
    cold.shade_list = shade_list;
    cold.partial = partial;
    cold.fc = fc;
    cold.n_pix = (uint32_t)a.n_pix;
    cold.chunk_spp = (uint32_t)a.chunk_spp;
    cold.spp = (uint32_t)a.spp;
    cold.items = sc.items;
    cold.n_items = sc.n_items;
    cold.n_pairs = sc.n_pairs;
  }
  __syncthreads();
  // the number of pixels that need sampling comes straight from the classification
  // kernel's counter: the host does not wait for it between the two launches
  const uint32_t n_shade = __atomic_load_n(&fc->n_shade, __ATOMIC_RELAXED);
  const uint32_t n_units = n_shade * (uint32_t)a.n_chunks;  // fits: the host bounds pixels * chunks
  // the wave's reservation: units [res_base, res_base + res_left) are not dealt yet
  uint32_t res_base = 0, res_left = 0;
  bool drained = false;  // the global cursor has passed the last unit
  uint32_t gpix = 0, k = 0;
  bool has_unit = false;
  // the path
  V3 pos = mk(0, 0, 0), N = pos, wi = pos, beta = pos, Lp = pos;
  uint32_t mat = 0, depth = 0;
  bool live = false;
  uint32_t n_vertices = 0, n_rays = 0;
  if constexpr (kLds) {
    us.n_vertices[tid] = 0;
    us.n_rays[tid] = 0;
  }

  SectionClock clk;
  clk.start();
  for (;;) {
    clk.mark(5);
    if (!live) {
      if (has_unit && k >= us.k_end[tid]) {
        const V3 acc = mk(us.acc[0][tid], us.acc[1][tid], us.acc[2][tid]);
        cold.partial[us.out[tid]] = pack(acc, 0.0f);
        has_unit = false;
      }
    }
    // deal units to the lanes that have none (wave-uniform control flow)
    for (unsigned long long want = __ballot(!live && !has_unit); want != 0;
         want = __ballot(!live && !has_unit)) {
      if (res_left == 0) {
        if (drained) break;
        unsigned long long base = 0;
        if (lane_id() == 0) base = atomicAdd(&cold.fc->next_unit, 64ull);
        base = __shfl(base, 0);
        if (base >= n_units) {
          drained = true;
          break;
        }
        res_base = (uint32_t)base;
        res_left = min(64u, n_units - res_base);
      }
      const uint32_t rank = (uint32_t)__popcll(want & lanes_below());
      if (!live && !has_unit && rank < res_left) {
        const uint32_t u = res_base + rank;
        const uint32_t entry = u % n_shade;
        const uint32_t chunk = u / n_shade;
        const uint32_t lp = cold.shade_list[entry];
        const float4 pm = cold.pc.pos_mat[lp], pn = cold.pc.nrm[lp], pw = cold.pc.wi[lp];
        us.fpos[0][tid] = pm.x; us.fpos[1][tid] = pm.y; us.fpos[2][tid] = pm.z;
        us.fmat[tid] = wbits(pm);
        us.fN[0][tid] = pn.x; us.fN[1][tid] = pn.y; us.fN[2][tid] = pn.z;
        us.fwi[0][tid] = pw.x; us.fwi[1][tid] = pw.y; us.fwi[2][tid] = pw.z;  // unit(wi) already
        gpix = wbits(pw);
        us.out[tid] = chunk * cold.n_pix + lp;  // below 2^32: checked by the host
        k = chunk * cold.chunk_spp;
        us.k_end[tid] = min(cold.spp, k + cold.chunk_spp);
        us.acc[0][tid] = 0; us.acc[1][tid] = 0; us.acc[2][tid] = 0;
        has_unit = true;
      }
      const uint32_t dealt = min((uint32_t)__popcll(want), res_left);
      res_base += dealt;
      res_left -= dealt;
    }
    if (!live) {
      if (has_unit) {
        // new sample: every sample of a pixel starts at the same first vertex
        pos = mk(us.fpos[0][tid], us.fpos[1][tid], us.fpos[2][tid]);
        N = mk(us.fN[0][tid], us.fN[1][tid], us.fN[2][tid]);
        wi = mk(us.fwi[0][tid], us.fwi[1][tid], us.fwi[2][tid]);
        mat = us.fmat[tid];
        beta = mk(1, 1, 1);
        if constexpr (kLds) { us.Lp[0][tid] = 0; us.Lp[1][tid] = 0; us.Lp[2][tid] = 0; }
        else Lp = mk(0, 0, 0);
        depth = 0;
        live = true;
      }
    }
    if (__ballot(live) == 0) break;
    clk.mark(0);
    clk.count(6);

    Emit e;
    e.has_shadow = false; e.has_ext = false;
    e.light_pos = mk(0, 0, 0); e.weight = mk(0, 0, 0); e.beta_next = mk(0, 0, 0);
    e.uo_ext = mk(0, 0, 0);
    e.light_pdf = 0;
    if (live) {
      // (wi holds unit(wi) in this kernel)
      if (kLds) shade_vertex<true>(a, LdsTables{lds, lsh, L.n_lights, L.total_area}, pos, N, wi, beta, mat, gpix, k, depth, e);
      else shade_vertex<true>(a, GlobalTables{sc, L, mats}, pos, N, wi, beta, mat, gpix, k, depth, e);
      const uint32_t new_rays = (e.has_shadow ? 1u : 0u) + (e.has_ext ? 1u : 0u);
      if constexpr (kLds) {
        us.n_vertices[tid] += 1u;
        us.n_rays[tid] += new_rays;
      } else {
        n_vertices++;
        n_rays += new_rays;
      }
    }
    // a lane without a shadow or an extension ray traces (1, 1, 1), whose normalisation is a constant:
    // 1 / RN(sqrt(3)), rounded (checked against unit() by rtr_self_test)
    if (!e.has_shadow) { e.l = mk(1, 1, 1); e.ln = mk(kUnitOfOnes, kUnitOfOnes, kUnitOfOnes); }
    if (!e.has_ext) { e.w1 = mk(1, 1, 1); e.wn = mk(kUnitOfOnes, kUnitOfOnes, kUnitOfOnes); }
    const V3 ln = e.ln, wn = e.wn;
    clk.mark(1);
    HitRec hs, he;
    int node_e = -1;
    if constexpr (kLds) {

      // both rays leave the same point: one pass over the leaves for the two of them
      const uint64_t table = (uint64_t)cold.items;
      const float4 *items = (const float4 *)(((uint64_t)wave_first((uint32_t)(table >> 32)) << 32) |
                                             wave_first((uint32_t)table));
      const HitPair hp = trace_all2<kBounded>(sc, items, (int)wave_first((uint32_t)cold.n_items),
                                    wave_first((uint32_t)cold.n_pairs) != 0u, lds, e.has_shadow, e.has_ext,
                                    pos, e.l, ln, e.w1, wn, ws.merge, clk);
      hs = hp.a;
      he = hp.b;
      node_e = hp.node_b;
    } else {
      hs = trace_any<kLds>(sc, lds, e.has_shadow, pos, e.l, ln, ws);
      he = trace_any<kLds>(sc, lds, e.has_ext, pos, e.w1, wn, ws);
    }
    if (live) {
      // next-event result: src/Render.cpp:95-107.  A miss fails the distance test
      // like the reference's `testInter.hit` does.
      if (e.has_shadow && hs.leaf >= 0) {
        V3 hp, hn;
        hit_point_any<kLds>(sc, lds, hs, pos, ln, hp, hn);
        if (near_light(hp - e.light_pos)) {
          V3 term = div_div(e.weight, hs.t * hs.t, e.light_pdf);
          const V3 add = beta * term;
          if constexpr (kLds) {
            us.Lp[0][tid] = us.Lp[0][tid] + add.x;
            us.Lp[1][tid] = us.Lp[1][tid] + add.y;
            us.Lp[2][tid] = us.Lp[2][tid] + add.z;
          } else {
            Lp = Lp + add;
          }
        }
      }
      // extension result: src/Render.cpp:117-125
      bool cont = false;
      if (e.has_ext && he.leaf >= 0) {
        int prim2, m2;
        bool emissive;
        hit_info<kLds>(sc, mats, lsh, he, node_e, prim2, m2, emissive);
        if (!emissive && !(a.max_depth > 0 && (int)depth + 1 >= a.max_depth)) {
          V3 hp, hn;
          hit_point_any<kLds>(sc, lds, he, pos, wn, hp, hn);
          beta = e.beta_next;
          wi = -e.uo_ext;
          N = unit_near(hn);  // a stored face normal or a sphere's normalised radius
          pos = hp;
          mat = (uint32_t)m2;
          depth++;
          cont = true;
        }
      }
      if (!cont) {
        // color = color + pathTracing(...), src/Render.cpp:659
        if constexpr (kLds) {
          us.acc[0][tid] = us.acc[0][tid] + us.Lp[0][tid];
          us.acc[1][tid] = us.acc[1][tid] + us.Lp[1][tid];
          us.acc[2][tid] = us.acc[2][tid] + us.Lp[2][tid];
        } else {
          us.acc[0][tid] = us.acc[0][tid] + Lp.x;
          us.acc[1][tid] = us.acc[1][tid] + Lp.y;
          us.acc[2][tid] = us.acc[2][tid] + Lp.z;
        }
        k++;
        live = false;
      }
    }
  }
  clk.flush();
  // statistics: one atomic pair per wave
  if constexpr (kLds) {
    n_vertices = us.n_vertices[tid];
    n_rays = us.n_rays[tid];
  }
  unsigned long long nv = n_vertices, nr = n_rays;
  for (int off = 32; off > 0; off >>= 1) {
    nv += __shfl_down(nv, off);
    nr += __shfl_down(nr, off);
  }
  if (lane_id() == 0) {
    atomicAdd(&fc->vertices, nv);
    atomicAdd(&fc->rays, nr);
  }
}

// ---- color / spp: src/Render.cpp:661-662 ----
__global__ void __launch_bounds__(kBlock)
k_resolve(RenderArgs a, const float4 *__restrict__ partial, float *__restrict__ rgb) {
  int lp = blockIdx.x * kBlock + threadIdx.x;
  if (lp >= a.n_pix) return;
  V3 c = mk(0, 0, 0);
  for (int ch = 0; ch < a.n_chunks; ch++) c = c + xyz(partial[(size_t)ch * a.n_pix + lp]);
  c = c / (float)a.spp;
  rgb[3 * (size_t)lp] = c.x;
  rgb[3 * (size_t)lp + 1] = c.y;
  rgb[3 * (size_t)lp + 2] = c.z;
}

// ---- photon emission: the thread function of Render::createPhotonMap, src/Render.cpp:407-463 ----
// The reference fills photonMap[k] one emission path after the other from the shared
// generator.  Here path number p draws bounce b's uniforms from the counter stream
// (pixel = p, sample = 0, bounce = b) under key (seed, 1), so paths are independent: one
// lane runs one path.  The map is the first photon_num photons of paths 0, 1, 2, ... in
// path order, which takes two passes over a batch of paths: the first counts what each
// path stores, a scan turns counts into positions, the second runs the same paths again
// and writes.  (A path that the reference would cut short because the map filled up
// mid-path is run to its end here; the photons past the end are not written.)
struct PhotonArgs {
  uint32_t seed, first_path, n_paths;
  unsigned long long base;   // photons stored by earlier batches
  unsigned long long limit;  // photon_num
};

// kPass 0: count what every path stores; 1: run the same paths again and write (the two passes of
// the fallback); 2: the usual way, one pass -- every photon goes to a staging array as it is found,
// in whatever order the waves reach it, tagged with its path and its number within the path, and
// k_photon_scatter moves it to its place once the paths' counts have been scanned.
enum { kPhotonCount = 0, kPhotonWrite = 1, kPhotonStage = 2 };
// Staging: every wave owns room for kPhotonWaveRoom photons (six per path; the default scene stores
// 1.8) and fills it front to back -- positions come from a counter the wave keeps to itself, so no
// memory round trip sits in the bounce loop.  A wave that runs out of room raises `overflow`: the
// batch is then run again the slow way.
constexpr unsigned kPhotonWaveRoom = 384;
struct PhotonStage {
  unsigned room;         // kPhotonWaveRoom (smaller in the test that drives a batch out of room)
  float4 *records;       // per wave kPhotonWaveRoom x 3: {pos, bits(path in batch)} {wi, bits(number in path)} {power, -}
  uint32_t *wave_count;  // photons each wave staged
  uint32_t *overflow;
};

template <bool kLds, int kPass>
__global__ void __launch_bounds__(kBlock)
k_photon_paths(PhotonArgs a, DeviceScene sc, DeviceLights L, const DeviceMaterial *__restrict__ mats,
               uint32_t *__restrict__ counts, unsigned long long *__restrict__ block_sums,
               const unsigned long long *__restrict__ block_offs, float *__restrict__ pos3,
               float *__restrict__ wi3, float *__restrict__ power3,
               unsigned int *__restrict__ paths_used, PhotonStage stage) {
  constexpr bool kWrite = kPass == kPhotonWrite;
  __shared__ typename WaveLds<kLds>::type wsh[kBlock / 64];
  __shared__ LdsScene lds;
  __shared__ unsigned s_cnt[kBlock];
  __shared__ unsigned s_total;
  if (kLds) stage_scene(sc, lds, threadIdx.x, kBlock);
  if (threadIdx.x == 0) s_total = 0;
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = p < a.n_paths;
  const uint32_t path = a.first_path + p;
  unsigned long long out = 0;
  if (kWrite) s_cnt[threadIdx.x] = valid ? counts[p] : 0u;
  __syncthreads();
  if (kWrite) {
    unsigned before = 0;
    for (unsigned t = 0; t < threadIdx.x; t++) before += s_cnt[t];
    out = a.base + block_offs[blockIdx.x] + before;
    // the path that stores photon number `limit` is the last one the reference starts
    if (valid && out < a.limit && out + s_cnt[threadIdx.x] >= a.limit) *paths_used = path + 1u;
  }
  const GlobalTables T{sc, L, mats};
  typename WaveLds<kLds>::type &ws = wsh[threadIdx.x >> 6];

  VertexStream rng(a.seed, path, 0u, 0u, 1u);
  bool active = valid;
  LightSample ls;
  ls.leaf = -1;
  ls.light = 0;
  ls.pos = mk(0, 0, 0);
  ls.nrm = mk(0, 0, 1);
  if (active) {
    sample_light(T, rng, ls);
    if (ls.leaf < 0) active = false;  // no light: the reference leaves its loop (:413-415)
  }
  V3 ppos = ls.pos, N = unit(ls.nrm), pwi = N, ppow = mk(0, 0, 0);
  DeviceMaterial im = mats[0];
  if (active) {
    const HitRec lh{0.0f, 0.0f, 0.0f, ls.leaf};
    im = mats[sc.prim_material[hit_prim(sc, lh)]];
    ppow = xyz(im.e_kind);
  }
  uint32_t k = 0, bounce = 0;
  unsigned staged = 0;  // kPhotonStage: photons the wave has put into its room (wave-uniform)
  int max_deep = 64;
  while (__ballot(active)) {
    V3 wo = mk(0, 0, 1);
    bool shoot = false;
    if (active) {
      max_deep--;
      if (bounce != 0) rng = VertexStream(a.seed, path, 0u, bounce, 1u);
      if (mat_kind(im) == 1) {
        wo = mat_mirror(pwi, N);
      } else {
        const float u1 = rng.next();
        const float u2 = rng.next();
        wo = mat_sample_uv(N, u1, u2);
      }
      // Vector3f::norm() is the squared length, compared with a double literal (:427)
      if ((double)(wo.x * wo.x + wo.y * wo.y + wo.z * wo.z) < 1e-5) {
        active = false;
      } else {
        wo = unit(wo);
        shoot = true;
      }
    }
    const V3 wn = unit(wo);
    const HitRec h = trace_any<kLds>(sc, lds, shoot, ppos, wo, wn, ws);
    bool store = false;
    V3 s_pos = mk(0, 0, 0), s_wi = s_pos, s_pow = s_pos;
    if (shoot) {
      bool go_on = false;
      if (h.leaf >= 0) {
        const DeviceMaterial nm = mats[sc.prim_material[hit_prim(sc, h)]];
        if (!mat_emissive(nm)) {
          V3 hp, hn;
          hit_point(sc, h, ppos, wn, hp, hn);
          const V3 nw = -wo;
          const V3 un = unit(hn);
          V3 npow;
          if (!mat_emissive(im)) npow = mat_eval(im, pwi, wo, N) * ppow * dot(N, wo) * dot(un, nw);
          else npow = ppow * dot(N, wo) * dot(un, nw);
          if (mat_kind(im) == 0) {  // stored only when the vertex left is DIFFUSE (:451)
            if (kWrite && out + k < a.limit) {
              const unsigned long long q = 3ull * (out + k);
              pos3[q] = hp.x; pos3[q + 1] = hp.y; pos3[q + 2] = hp.z;
              wi3[q] = nw.x; wi3[q + 1] = nw.y; wi3[q + 2] = nw.z;
              power3[q] = npow.x; power3[q + 1] = npow.y; power3[q + 2] = npow.z;
            }
            if (kPass == kPhotonStage) {
              store = true;
              s_pos = hp; s_wi = nw; s_pow = npow;
            } else {
              k++;
            }
          }
          ppos = hp;
          pwi = nw;
          ppow = npow;
          N = un;
          im = nm;
          go_on = true;
        }
      }
      if (!go_on) {
        active = false;
      } else if (max_deep < 0) {  // :461; `k < endIndex` is the write limit above
        active = false;
      } else {
        const float u = rng.next();
        if (!(u < 0.6f || mat_kind(im) != 0)) active = false;
        bounce++;
      }
    }
    if (kPass == kPhotonStage) {
      // the wave's photons of this bounce, appended to the wave's own room by rank in the ballot
      const unsigned long long m = __ballot(store);
      if (store) {
        const unsigned at = staged + (unsigned)__popcll(m & lanes_below());
        if (at < stage.room) {
          float4 *rec = stage.records + 3 * ((size_t)(p >> 6) * stage.room + at);
          rec[0] = pack(s_pos, p);
          rec[1] = pack(s_wi, k);
          rec[2] = pack(s_pow, 0u);
        }
        k++;
      }
      staged += (unsigned)__popcll(m);
    }
  }
  if (kPass == kPhotonStage && lane_id() == 0) {
    stage.wave_count[p >> 6] = staged < stage.room ? staged : stage.room;
    if (staged > stage.room) *stage.overflow = 1u;
  }
  if (!kWrite) {
    if (valid) counts[p] = k;
    if (k) atomicAdd(&s_total, k);
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_total;
  }
}

// exclusive scan of the per-workgroup photon counts of a batch (one workgroup)
__global__ void __launch_bounds__(kBlock)
k_photon_scan(const unsigned long long *__restrict__ block_sums, unsigned long long *__restrict__ block_offs,
              unsigned n_blocks, unsigned long long *__restrict__ total) {
  __shared__ unsigned long long s_part[kBlock];
  const unsigned per = (n_blocks + kBlock - 1) / kBlock;
  const unsigned b0 = threadIdx.x * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  unsigned long long sum = 0;
  for (unsigned b = b0; b < b1; b++) sum += block_sums[b];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  unsigned long long before = 0;
  for (unsigned t = 0; t < threadIdx.x; t++) before += s_part[t];
  for (unsigned b = b0; b < b1; b++) {
    block_offs[b] = before;
    before += block_sums[b];
  }
  if (threadIdx.x == kBlock - 1) *total = before;
}

// Every path's first position in the map, from the paths' counts and the workgroups' offsets; the
// path that stores photon number `limit` is the last one the reference starts.
__global__ void __launch_bounds__(kBlock)
k_photon_offsets(PhotonArgs a, const uint32_t *__restrict__ counts, const unsigned long long *__restrict__ block_offs,
                 unsigned long long *__restrict__ path_first, unsigned int *__restrict__ paths_used) {
  __shared__ unsigned s_cnt[kBlock];
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  const unsigned mine = p < a.n_paths ? counts[p] : 0u;
  s_cnt[threadIdx.x] = mine;
  __syncthreads();
  // inclusive scan over the workgroup
  for (unsigned d = 1; d < kBlock; d <<= 1) {
    const unsigned add = threadIdx.x >= d ? s_cnt[threadIdx.x - d] : 0u;
    __syncthreads();
    s_cnt[threadIdx.x] += add;
    __syncthreads();
  }
  if (p >= a.n_paths) return;
  const unsigned long long first = a.base + block_offs[blockIdx.x] + (s_cnt[threadIdx.x] - mine);
  path_first[p] = first;
  if (first < a.limit && first + mine >= a.limit) *paths_used = a.first_path + p + 1u;
}

// staged photons to their places: path order, and within a path the order they were found in
__global__ void __launch_bounds__(kBlock)
k_photon_scatter(PhotonArgs a, PhotonStage stage, const unsigned long long *__restrict__ path_first,
                 float *__restrict__ pos3, float *__restrict__ wi3, float *__restrict__ power3) {
  const unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long wave = i / stage.room;
  if (wave >= (a.n_paths + 63u) / 64u || i % stage.room >= stage.wave_count[wave]) return;
  const float4 r0 = stage.records[3 * i], r1 = stage.records[3 * i + 1], r2 = stage.records[3 * i + 2];
  const unsigned long long at = path_first[wbits(r0)] + wbits(r1);
  if (at >= a.limit) return;
  const unsigned long long q = 3ull * at;
  pos3[q] = r0.x; pos3[q + 1] = r0.y; pos3[q + 2] = r0.z;
  wi3[q] = r1.x; wi3[q + 1] = r1.y; wi3[q + 2] = r1.z;
  power3[q] = r2.x; power3[q + 1] = r2.y; power3[q + 2] = r2.z;
}

// ---- unit-level entry points used by the parity tests ----
struct ArrayStream {
  const float *u;
  uint32_t dim;
  uint32_t n;
  RTR_DEV float next() {
    float v = dim < n ? u[dim] : 0.0f;
    dim++;
    return v;
  }
};

__global__ void __launch_bounds__(kBlock)
k_material_eval(const DeviceMaterial *__restrict__ mats, const int32_t *__restrict__ mid,
                const float *__restrict__ wi, const float *__restrict__ wo,
                const float *__restrict__ nn, size_t n, float *__restrict__ f,
                float *__restrict__ pdf) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  DeviceMaterial dm = mats[mid[i]];
  V3 r = mat_eval(dm, mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]),
                  mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]),
                  mk(nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]));
  if (f) { f[3 * i] = r.x; f[3 * i + 1] = r.y; f[3 * i + 2] = r.z; }
  if (pdf) pdf[i] = mat_pdf(dm);
}

__global__ void __launch_bounds__(kBlock)
k_material_sample(const DeviceMaterial *__restrict__ mats, const int32_t *__restrict__ mid,
                  const float *__restrict__ wi, const float *__restrict__ nn,
                  const float *__restrict__ u2, size_t n, float *__restrict__ wo) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  DeviceMaterial dm = mats[mid[i]];
  V3 w = mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
  V3 N = mk(nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]);
  V3 r = mat_kind(dm) == 1 ? mat_mirror(w, N) : mat_sample_uv(N, u2[2 * i], u2[2 * i + 1]);
  wo[3 * i] = r.x; wo[3 * i + 1] = r.y; wo[3 * i + 2] = r.z;
}

__global__ void __launch_bounds__(kBlock)
k_sample_light(DeviceScene sc, DeviceLights L, const float *__restrict__ u, int stride, size_t n,
               float *__restrict__ pos, float *__restrict__ nrm, float *__restrict__ pdf,
               int32_t *__restrict__ prim, int32_t *__restrict__ used) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  ArrayStream rng{u + (size_t)stride * i, 0u, (uint32_t)stride};
  LightSample ls;
  sample_light(GlobalTables{sc, L, nullptr}, rng, ls);
  pos[3 * i] = ls.pos.x; pos[3 * i + 1] = ls.pos.y; pos[3 * i + 2] = ls.pos.z;
  nrm[3 * i] = ls.nrm.x; nrm[3 * i + 1] = ls.nrm.y; nrm[3 * i + 2] = ls.nrm.z;
  pdf[i] = ls.pdf;
  prim[i] = ls.leaf < 0 ? -1
                        : ((ls.leaf >> 30) ? sc.sph_prim[ls.leaf & 0x3fffffff] : sc.tri_prim[ls.leaf]);
  used[i] = (int32_t)rng.dim;
}

__global__ void __launch_bounds__(kBlock)
k_stream_uniform(uint32_t seed, const uint32_t *__restrict__ pixel,
                 const uint32_t *__restrict__ sample, const uint32_t *__restrict__ bounce,
                 const uint32_t *__restrict__ dim, size_t n, float *__restrict__ u) {
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  VertexStream rng(seed, pixel[i], sample[i], bounce[i]);
  rng.dim = dim[i];
  u[i] = rng.next();
}

} // namespace rtr
