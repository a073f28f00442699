// librtr_hip.so: the C ABI declared in include/rtr.h over the HIP kernels.
// gfx950 only; there is no CPU implementation behind any of these entry points.
#include "../../include/rtr.h"
#include "rtr_kernels.hpp"
#include "rtr_build.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace rtr;

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg) {
  g_error = msg;
  return code;
}

}  // namespace

namespace rtr {
// for the library's other translation units: sets what rtr_last_error() returns
int fail_msg(int code, const std::string &msg) { return fail(code, msg); }
}  // namespace rtr

namespace {

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(RTR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

template <class T>
struct DevBuf {
  T *ptr = nullptr;
  size_t cap = 0;  // elements
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&ptr, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
};

struct Workspace {
  DevBuf<float4> pix_dir, pix_hit, pc_pos, pc_nrm, pc_wi, partial;
  DevBuf<float4> pool[14];
  DevBuf<uint32_t> shade_list, queue;
  DevBuf<uint2> wave_res;  // wavefront loop: each wave's reserved, not yet dealt units
  DevBuf<float> rgb;
  DevBuf<float> gather;  // multi-device render: every partition's framebuffer, one after the other
  DevBuf<Counters> counters;
  DevBuf<BlockCounters> block_counters;
  Counters *host_counters = nullptr;             // pinned
  BlockCounters *host_block_counters = nullptr;  // pinned
  size_t host_block_cap = 0;
  float *host_rgb = nullptr;  // pinned: the framebuffer on its way to the caller's array
  size_t host_rgb_cap = 0;
  // A copy from the device straight into pageable memory goes through the runtime's own small
  // staging buffers (2.4 ms for a 784 x 784 image); into pinned memory it runs at link speed,
  // and the copy from there to the caller's array is a plain memcpy (0.9 ms together).
  hipError_t reserve_host_rgb(size_t n) {
    if (n <= host_rgb_cap) return hipSuccess;
    if (host_rgb) (void)hipHostFree(host_rgb);
    host_rgb = nullptr;
    host_rgb_cap = 0;
    hipError_t e = hipHostMalloc((void **)&host_rgb, n * sizeof(float));
    if (e == hipSuccess) host_rgb_cap = n;
    return e;
  }
  void release() {
    pix_dir.release(); pix_hit.release(); pc_pos.release(); pc_nrm.release(); pc_wi.release();
    partial.release();
    for (auto &b : pool) b.release();
    shade_list.release(); queue.release(); rgb.release(); gather.release(); counters.release();
    wave_res.release();
    block_counters.release();
    if (host_counters) (void)hipHostFree(host_counters);
    if (host_block_counters) (void)hipHostFree(host_block_counters);
    if (host_rgb) (void)hipHostFree(host_rgb);
    host_counters = nullptr;
    host_block_counters = nullptr;
    host_block_cap = 0;
    host_rgb = nullptr;
    host_rgb_cap = 0;
  }
};

}  // namespace

struct rtr_scene {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // second half of the path pool runs here, overlapping the first
  DevBuf<float4> nodes, tris, sphs, lights, leaves, items, pair_second;
  DevBuf<int32_t> parent, node_info;
  DevBuf<float4> light_le;
  int n_light_tris = 0;
  DevBuf<int32_t> tri_prim, sph_prim, prim_kind_row, prim_material, light_tris;
  DevBuf<float> tri_area;
  DevBuf<DeviceMaterial> mats;
  DeviceScene ds{};
  DeviceLights dl{};
  int n_materials = 0;
  bool lds_geometry = false;  // scene small enough for the kernels to stage it in LDS
  int fused_grid = 0;         // resident workgroups of the fused sampling kernel (0: not queried yet)
  unsigned char *noalloc_scratch = nullptr;  // this device's copy of the scratch array
  int stream_grid = 0;        // the same for k_trace_api_stream
  int queue_grid = 0;         // and for k_trace_queue_stream
  int connect_grid = 0;       // and for k_check_connectable_stream
  // ray counters of the streaming traversal, used in rotation so that calls queued on
  // different streams do not share one
  DevBuf<unsigned long long> trace_next;
  unsigned trace_next_turn = 0;
  // one event per counter, recorded behind the launch that uses it: a counter is handed
  // out again only once that launch has finished (the call waits if it has not)
  hipEvent_t trace_done[64] = {};
  bool trace_busy[64] = {};
  // photon emission: per-path counts of a batch, per-workgroup sums and offsets,
  // {photons in the batch, -}, number of paths the map took
  DevBuf<uint32_t> ph_counts, ph_paths_used;
  DevBuf<unsigned long long> ph_sums, ph_offs, ph_total, ph_first;
  DevBuf<uint32_t> ph_wave_count;  // per wave of a batch, and the overflow flag after them
  DevBuf<float4> ph_stage;  // photons of a batch before they are put in path order
  Workspace ws;
  // scratch for the host-pointer entry points
  DevBuf<float> io_f[5];
  DevBuf<int32_t> io_i[2];
};

namespace {

constexpr unsigned kTraceCounters = 64;  // size of rtr_scene::trace_done
constexpr int kFusedMaxNodes = 1 << 19;  // RTR_MODE_AUTO: largest tree the fused kernel is chosen for

struct DeviceGuard {
  int prev = 0;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (ok) (void)hipSetDevice(prev);
  }
};

template <class T>
int upload(DevBuf<T> &buf, const std::vector<T> &host) {
  HIP_TRY(buf.reserve(host.size() ? host.size() : 1));
  if (!host.empty())
    HIP_TRY(hipMemcpy(buf.ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return RTR_OK;
}

unsigned blocks_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// Resident workgroups of a persistent kernel on the scene's device.
template <class Kernel>
int resident_grid(rtr_scene *s, Kernel kernel, int &cached_grid) {
  if (cached_grid == 0) {
    int per_cu = 0, cus = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
    if (per_cu < 1 || cus < 1) return fail(RTR_ERR_DEVICE, "occupancy query failed");
    cached_grid = per_cu * cus;
  }
  return RTR_OK;
}

// Launch shape and ray counter of a streaming traversal kernel over n rays: as many
// workgroups as stay resident (never more than n needs), and the next of the scene's
// rotating counters, zeroed on `stream` ahead of the launch.  The counter's previous
// launch (kTraceCounters calls ago, possibly on another stream) must be over before it
// is zeroed again: its event says so, and the call waits for it otherwise.  After the
// launch the caller records the event again with stream_launched().
template <class Kernel>
int stream_launch(rtr_scene *s, Kernel kernel, int &cached_grid, size_t n, hipStream_t stream,
                  unsigned &grid, unsigned long long *&next, unsigned &turn) {
  if (int rc = resident_grid(s, kernel, cached_grid)) return rc;
  HIP_TRY(s->trace_next.reserve(kTraceCounters));
  turn = s->trace_next_turn++ % kTraceCounters;
  if (!s->trace_done[turn])
    HIP_TRY(hipEventCreateWithFlags(&s->trace_done[turn], hipEventDisableTiming));
  if (s->trace_busy[turn]) {
    HIP_TRY(hipEventSynchronize(s->trace_done[turn]));
    s->trace_busy[turn] = false;
  }
  next = s->trace_next.ptr + turn;
  HIP_TRY(hipMemsetAsync(next, 0, sizeof(*next), stream));
  grid = (unsigned)cached_grid;
  if ((uint64_t)grid * kBlock > n) grid = blocks_for(n);
  return RTR_OK;
}

int stream_launched(rtr_scene *s, unsigned turn, hipStream_t stream) {
  HIP_TRY(hipEventRecord(s->trace_done[turn], stream));
  s->trace_busy[turn] = true;
  return RTR_OK;
}

// The description's tree and primitives -> the scene's device arrays, by the kernels of
// rtr_build.hpp: the host only copies the caller's arrays over and reads back what went wrong.
int build_scene_arrays(rtr_scene *s, const rtr_scene_desc *d, bool info_fits, BuildFlags &flags,
                       int32_t &n_leaves) {
  const size_t N = (size_t)d->n_nodes, P = (size_t)d->n_prims, T = (size_t)d->n_tris,
               S = (size_t)d->n_spheres;
  hipStream_t st = s->stream;
  // leaves are counted here because their records are allocated before the tree is checked
  size_t L = 0;
  for (size_t i = 0; i < N; i++) L += (d->node_left[i] < 0 && d->node_right[i] < 0) ? 1 : 0;
  n_leaves = (int32_t)L;

  // one staging buffer for all input arrays (4-byte elements), freed on return
  struct Piece { const void *host; size_t count; size_t offset; };
  std::vector<Piece> pieces;
  size_t total = 0;
  auto add = [&](const void *host, size_t count) {
    pieces.push_back(Piece{host, count, total});
    total += (count + 63) / 64 * 64;  // keep every array 256-byte aligned
    return pieces.size() - 1;
  };
  size_t i_bmin[3], i_bmax[3], i_p0[3], i_p1[3], i_p2[3], i_tn[3], i_sc[3];
  for (int a = 0; a < 3; a++) { i_bmin[a] = add(d->node_bmin[a], N); i_bmax[a] = add(d->node_bmax[a], N); }
  const size_t i_left = add(d->node_left, N), i_right = add(d->node_right, N), i_prim = add(d->node_prim, N);
  const size_t i_pk = add(d->prim_kind, P), i_pi = add(d->prim_index, P), i_pm = add(d->prim_material, P);
  for (int a = 0; a < 3; a++) {
    i_p0[a] = add(d->tri_p0[a], T); i_p1[a] = add(d->tri_p1[a], T);
    i_p2[a] = add(d->tri_p2[a], T); i_tn[a] = add(d->tri_n[a], T);
  }
  const size_t i_ta = add(d->tri_area, T);
  for (int a = 0; a < 3; a++) i_sc[a] = add(d->sph_c[a], S);
  const size_t i_sr = add(d->sph_r, S);
  DevBuf<uint32_t> stage, temp;
  DevBuf<BuildFlags> dflags;
  struct Cleanup {
    DevBuf<uint32_t> &a, &b;
    DevBuf<BuildFlags> &c;
    ~Cleanup() { a.release(); b.release(); c.release(); }
  } cleanup{stage, temp, dflags};
  HIP_TRY(stage.reserve(total ? total : 1));
  for (const Piece &p : pieces)
    if (p.count)
      HIP_TRY(hipMemcpyAsync(stage.ptr + p.offset, p.host, 4 * p.count, hipMemcpyHostToDevice, st));
  auto fptr = [&](size_t k) { return (const float *)(stage.ptr + pieces[k].offset); };
  auto iptr = [&](size_t k) { return (const int32_t *)(stage.ptr + pieces[k].offset); };
  BuildInput in{};
  for (int a = 0; a < 3; a++) {
    in.bmin[a] = fptr(i_bmin[a]); in.bmax[a] = fptr(i_bmax[a]);
    in.p0[a] = fptr(i_p0[a]); in.p1[a] = fptr(i_p1[a]); in.p2[a] = fptr(i_p2[a]); in.tn[a] = fptr(i_tn[a]);
    in.sc[a] = fptr(i_sc[a]);
  }
  in.left = iptr(i_left); in.right = iptr(i_right); in.prim = iptr(i_prim);
  in.prim_kind = iptr(i_pk); in.prim_index = iptr(i_pi); in.prim_material = iptr(i_pm);
  in.tri_area = fptr(i_ta); in.sr = fptr(i_sr);
  in.n_nodes = d->n_nodes; in.root = d->root; in.n_prims = d->n_prims; in.n_tris = d->n_tris;
  in.n_sphs = d->n_spheres; in.n_materials = d->n_materials;

  HIP_TRY(temp.reserve(3 * N + 1));
  BuildTemp tmp{(int32_t *)temp.ptr, temp.ptr + N, (int32_t *)(temp.ptr + 2 * N)};
  HIP_TRY(dflags.reserve(1));
  HIP_TRY(s->nodes.reserve(N ? 2 * N : 1));
  HIP_TRY(s->parent.reserve(N ? N : 1));
  HIP_TRY(s->node_info.reserve(N ? N : 1));
  HIP_TRY(s->leaves.reserve(4 * (L + 1)));  // + the padding record trace_all loads past the end
  HIP_TRY(s->tris.reserve(T ? 3 * T : 1));
  HIP_TRY(s->sphs.reserve(S ? S : 1));
  HIP_TRY(s->tri_prim.reserve(T ? T : 1));
  HIP_TRY(s->sph_prim.reserve(S ? S : 1));
  HIP_TRY(s->tri_area.reserve(T ? T : 1));
  HIP_TRY(s->prim_kind_row.reserve(P ? P : 1));
  HIP_TRY(s->prim_material.reserve(P ? P : 1));
  BuildOutput out{s->nodes.ptr, s->tris.ptr, s->sphs.ptr, s->leaves.ptr, s->parent.ptr, s->node_info.ptr,
                  s->tri_prim.ptr, s->sph_prim.ptr, s->prim_kind_row.ptr, s->prim_material.ptr,
                  s->tri_area.ptr, s->mats.ptr, info_fits ? 1 : 0};
  HIP_TRY(hipMemsetAsync(dflags.ptr, 0, sizeof(BuildFlags), st));
  HIP_TRY(hipMemsetAsync(s->leaves.ptr, 0, 4 * (L + 1) * sizeof(float4), st));
  if (T) HIP_TRY(hipMemsetAsync(s->tri_prim.ptr, 0xff, T * 4, st));
  if (S) HIP_TRY(hipMemsetAsync(s->sph_prim.ptr, 0xff, S * 4, st));
  if (N) {
    HIP_TRY(hipMemsetAsync(tmp.parent_old, 0xff, N * 4, st));
    HIP_TRY(hipMemsetAsync(tmp.leaves_under, 0, N * 4, st));
  }
  auto grid = [](size_t n) { return dim3((unsigned)((n + kBuildBlock - 1) / kBuildBlock)); };
  if (P) hipLaunchKernelGGL(k_build_prims, grid(P), dim3(kBuildBlock), 0, st, in, out, dflags.ptr);
  if (T || S)
    hipLaunchKernelGGL(k_build_geometry, grid(std::max(T, S)), dim3(kBuildBlock), 0, st, in, out, dflags.ptr);
  if (N) {
    hipLaunchKernelGGL(k_build_links, grid(N), dim3(kBuildBlock), 0, st, in, tmp, dflags.ptr);
    hipLaunchKernelGGL(k_build_count, grid(N), dim3(kBuildBlock), 0, st, in, tmp, dflags.ptr);
  }
  HIP_TRY(hipGetLastError());
  uint32_t under_root = 0;
  HIP_TRY(hipMemcpyAsync(&flags, dflags.ptr, sizeof(flags), hipMemcpyDeviceToHost, st));
  if (N) HIP_TRY(hipMemcpyAsync(&under_root, tmp.leaves_under + d->root, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const struct { uint32_t bit; const char *what; } kinds[] = {
      {kBuildPrimKind, "unknown prim_kind"},
      {kBuildPrimIndex, "prim_index out of range"},
      {kBuildPrimMaterial, "prim_material out of range"},
      {kBuildChildRange, "child index out of range"},
      {kBuildOneChild, "inner node with one child"},
      {kBuildLeafPrim, "leaf without a valid prim"},
      {kBuildTwoParents, "node reachable twice (not a tree)"},
      {kBuildTooDeep, "tree deeper than 64 levels"},
      {kBuildUnreachable, "unreachable nodes in the tree"}};
  for (const auto &k : kinds)
    if (flags.errors & k.bit) return fail(RTR_ERR_INVALID, k.what);
  // a proper tree with L leaves has 2L - 1 nodes, all of them under the root
  if (N && (under_root != L || N != 2 * L - 1)) return fail(RTR_ERR_INVALID, "unreachable nodes in the tree");
  if (N) {
    hipLaunchKernelGGL(k_build_place, grid(N), dim3(kBuildBlock), 0, st, in, tmp, out, dflags.ptr);
    hipLaunchKernelGGL(k_build_parent, grid(N), dim3(kBuildBlock), 0, st, in, tmp, out, dflags.ptr);
    if (L <= (size_t)kLdsLeaves) {
      HIP_TRY(s->items.reserve(4 * (L + 1)));
      HIP_TRY(s->pair_second.reserve(3 * (L / 2 + 1)));
      hipLaunchKernelGGL(k_build_items, dim3(1), dim3(64), 0, st, s->leaves.ptr, (int)L, out.tris, s->items.ptr,
                         s->pair_second.ptr, dflags.ptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&flags, dflags.ptr, sizeof(flags), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  for (const auto &k : kinds)
    if (flags.errors & k.bit) return fail(RTR_ERR_INVALID, k.what);
  return RTR_OK;
}

// src/Render.cpp:639-649: the camera basis and per-pixel steps, in the
// reference's mixed float/double arithmetic.
CameraSetup camera_setup(const rtr_camera &c, int H) {
  const double pi = 3.1415926;
  auto v = [](const float *p) { return V3{p[0], p[1], p[2]}; };
  V3 front = v(c.front), up = v(c.up);
  V3 cr = V3{front.y * up.z - front.z * up.y, front.z * up.x - up.z * front.x,
             front.x * up.y - front.y * up.x};
  float len = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
  V3 right = V3{cr.x / len, cr.y / len, cr.z / len};
  double scale = (double)tanf((float)((double)c.fov / 2.0 / pi / 2.0)) * 2.0 / (double)(float)H;
  float s = (float)scale;
  CameraSetup cs;
  cs.eye = v(c.position);
  cs.front = front;
  cs.vstep = V3{s * up.x, s * up.y, s * up.z};
  cs.hstep = V3{s * right.x, s * right.y, s * right.z};
  return cs;
}

void partition_columns(int W, int part_index, int part_count, int tile_cols,
                       std::vector<int32_t> &cols) {
  cols.clear();
  if (part_count <= 1) {
    for (int i = 0; i < W; i++) cols.push_back(i);
    return;
  }
  if (tile_cols <= 0) tile_cols = 8;
  for (int i = 0; i < W; i++)
    if ((i / tile_cols) % part_count == part_index) cols.push_back(i);
}

int check_xyz(const float *const p[3], const char *name) {
  if (!p[0] || !p[1] || !p[2]) return fail(RTR_ERR_INVALID, std::string(name) + " has a null axis");
  return RTR_OK;
}

// Number of image columns partition part_index of part_count renders (see ColumnMap).
int count_columns(int W, int part_index, int part_count, int tile_cols) {
  if (part_count <= 1) return W;
  const int n_tiles = (W + tile_cols - 1) / tile_cols;
  if (part_index >= n_tiles) return 0;
  const int mine = (n_tiles - part_index + part_count - 1) / part_count;
  int n = mine * tile_cols;
  if ((n_tiles - 1) % part_count == part_index) n -= n_tiles * tile_cols - W;  // the narrow last tile
  return n;
}

// Widest tile of at most 8 columns that deals every partition the same number of columns;
// 8 if there is none.
int balanced_tile_cols(int W, int part_count) {
  if (part_count <= 1) return 8;
  for (int t = 8; t >= 1; t--)
    if (W % t == 0 && (W / t) % part_count == 0) return t;
  return 8;
}

// Event pairs around kernels (rtr_render_params.profile); destroyed with the object, so
// no return path of a render leaks them.
struct EventSpans {
  struct Span { hipEvent_t a, b; int kind; bool ok; };
  bool on = false;
  std::vector<Span> live, idle;
  double ms[4] = {0, 0, 0, 0};  // 0 trace, 1 logic, 2 other, 3 fused
  ~EventSpans() {
    for (auto *v : {&live, &idle})
      for (Span &sp : *v) {
        if (sp.a) (void)hipEventDestroy(sp.a);
        if (sp.b) (void)hipEventDestroy(sp.b);
      }
  }
  // Only after the stream the spans were recorded on has been synchronised.
  void harvest() {
    for (Span &sp : live) {
      float t = 0;
      if (sp.ok && hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) ms[sp.kind] += t;
      idle.push_back(sp);
    }
    live.clear();
  }
  template <class Launch>
  hipError_t run(hipStream_t st, int kind, Launch &&launch) {
    Span sp{nullptr, nullptr, kind, false};
    if (on) {
      if (!idle.empty()) {
        sp = idle.back();
        idle.pop_back();
        sp.kind = kind;
      } else {
        hipError_t e = hipEventCreate(&sp.a);
        if (e == hipSuccess) e = hipEventCreate(&sp.b);
        if (e != hipSuccess) {
          if (sp.a) (void)hipEventDestroy(sp.a);
          return e;
        }
      }
      sp.ok = hipEventRecord(sp.a, st) == hipSuccess;
    }
    launch();
    hipError_t e = hipGetLastError();
    if (on) {
      sp.ok = (hipEventRecord(sp.b, st) == hipSuccess) && sp.ok;  // a failed record: the span is not counted
      live.push_back(sp);
    }
    return e;
  }
};
constexpr int kSpanTrace = 0, kSpanLogic = 1, kSpanOther = 2, kSpanFused = 3;

// What a render call asks for, checked and with the defaults filled in.
struct RenderPlan {
  RenderArgs a;
  ColumnMap cm;
  CameraSetup cs;
  size_t units;  // n_pix * n_chunks: partial sums
  bool fused;
};

// Checks shared by every render entry point; fills `plan`.  unit_budget bounds
// n_pix * n_chunks when the caller leaves chunk_spp to the library.
int plan_render(const rtr_scene *s, const rtr_camera *cam, const rtr_render_params *prm,
                size_t unit_budget, RenderPlan &plan) {
  if (prm->width <= 0 || prm->height <= 0 || prm->spp <= 0)
    return fail(RTR_ERR_INVALID, "width, height and spp must be positive");
  if ((int64_t)prm->width * prm->height > 0x7fffffff) return fail(RTR_ERR_INVALID, "image too large");
  if (prm->part_count > 1 && (prm->part_index < 0 || prm->part_index >= prm->part_count))
    return fail(RTR_ERR_INVALID, "part_index out of range");
  if (prm->mode < 0 || prm->mode > 2) return fail(RTR_ERR_INVALID, "mode must be 0, 1 or 2");
  RenderArgs &a = plan.a;
  a = RenderArgs{};
  plan.cm.part_count = prm->part_count > 1 ? prm->part_count : 1;
  plan.cm.part_index = prm->part_count > 1 ? prm->part_index : 0;
  plan.cm.tile_cols = prm->tile_cols > 0 ? prm->tile_cols : 8;
  a.W = prm->width; a.H = prm->height; a.spp = prm->spp;
  a.n_cols = count_columns(a.W, plan.cm.part_index, plan.cm.part_count, plan.cm.tile_cols);
  a.n_pix = a.n_cols * a.H;
  a.seed = prm->seed;
  a.max_depth = prm->max_depth > 0 ? prm->max_depth : 0;
  a.rr = prm->rr_prob > 0 ? prm->rr_prob : 0.6f;
  if (prm->chunk_spp > 0) {
    a.chunk_spp = prm->chunk_spp;
  } else {
    // default: up to 128 partial sums per pixel -- units small enough that the last one a
    // lane works on is a small share of its run, also when the image is split over 8
    // GPUs -- but no more units than the budget (large images get fewer, longer chunks)
    // (sized by the whole image, not by this partition: every partition of an image must
    // sum its samples in the same order as a one-device render of it)
    size_t chunks = unit_budget / ((size_t)a.W * (size_t)a.H);
    chunks = std::min<size_t>(std::max<size_t>(chunks, 1), 128);
    a.chunk_spp = (int)((prm->spp + chunks - 1) / chunks);
  }
  if (a.chunk_spp > a.spp) a.chunk_spp = a.spp;
  a.n_chunks = (a.spp + a.chunk_spp - 1) / a.chunk_spp;
  plan.units = (size_t)a.n_pix * a.n_chunks;
  if (plan.units > 0xfffffff0ull) return fail(RTR_ERR_INVALID, "too many work units; raise chunk_spp");
  plan.cs = camera_setup(*cam, a.H);
  // one persistent kernel with path state in registers, or the wavefront loop over a pool
  // of paths in HBM (measured on height-field meshes, fused kernel against wavefront loop,
  // Msamples/s: 6 K nodes 1255 : 708 and 360 K nodes 674 : 545 at 64 spp; 160 K nodes
  // 811 : 735, 360 K nodes 691 : 672 and 2 M nodes 401 : 460 at 256 spp)
  plan.fused = prm->mode == RTR_MODE_FUSED ||
               (prm->mode == RTR_MODE_AUTO && (s->lds_geometry || s->ds.n_nodes <= kFusedMaxNodes));
  return RTR_OK;
}

// Device memory every render needs, whichever way the sampling loop runs.
struct RenderBuffers {
  float4 *pix_dir, *pix_hit, *partial;
  PixelCache pc;
  uint32_t *shade_list;
  Counters *cnt;
};

// ray generation, primary rays, first vertex of every pixel
int enqueue_front(rtr_scene *s, const RenderPlan &pl, const RenderBuffers &b, hipStream_t st,
                  EventSpans &ev) {
  const RenderArgs &a = pl.a;
  HIP_TRY(hipMemsetAsync(b.cnt, 0, sizeof(Counters), st));
  HIP_TRY(ev.run(st, kSpanOther, [&] {
    hipLaunchKernelGGL(k_raygen, dim3(blocks_for(a.n_pix)), dim3(kBlock), 0, st, a, pl.cs, pl.cm, b.pix_dir);
  }));
  HIP_TRY(ev.run(st, kSpanTrace, [&] {
    if (s->lds_geometry)
      hipLaunchKernelGGL(k_trace_primary<true>, dim3(blocks_for(a.n_pix)), dim3(kBlock), 0, st, s->ds,
                         pl.cs, b.pix_dir, a.n_pix, b.pix_hit);
    else
      hipLaunchKernelGGL(k_trace_primary<false>, dim3(blocks_for(a.n_pix)), dim3(kBlock), 0, st, s->ds,
                         pl.cs, b.pix_dir, a.n_pix, b.pix_hit);
  }));
  HIP_TRY(ev.run(st, kSpanOther, [&] {
    hipLaunchKernelGGL(k_classify, dim3(blocks_for(a.n_pix)), dim3(kBlock), 0, st, a, s->ds, s->mats.ptr,
                       pl.cs, b.pix_dir, b.pix_hit, b.pc, b.shade_list, b.cnt, b.partial);
  }));
  return RTR_OK;
}

// The flavour of the fused kernel that takes the short cuts of a scene whose arithmetic cannot
// overflow: planes in the division-free test's range, nested boxes, primitives within 2^20.
bool bounded_scene(const rtr_scene *s) {
  return s->lds_geometry && s->ds.fast_planes != 0 && s->ds.nested_boxes != 0 && s->ds.near_prims != 0;
}

int fused_grid(rtr_scene *s) {
  if (bounded_scene(s)) return resident_grid(s, k_render_fused<true, true>, s->fused_grid);
  if (s->lds_geometry) return resident_grid(s, k_render_fused<true, false>, s->fused_grid);
  return resident_grid(s, k_render_fused<false, false>, s->fused_grid);
}

// The whole sampling loop as one launch.  The kernel reads the number of pixels that need
// sampling from the classification kernel's counter on the device, so nothing waits
// between the two.  s->fused_grid must be known (fused_grid()).
int enqueue_fused(rtr_scene *s, const RenderPlan &pl, const RenderBuffers &b, hipStream_t st,
                  EventSpans &ev) {
  const RenderArgs &a = pl.a;
  // as many workgroups as the device keeps resident at once; every lane then works
  // through its share of the units
  unsigned grid = (unsigned)s->fused_grid;
  if ((uint64_t)grid * kBlock > pl.units) grid = (unsigned)((pl.units + kBlock - 1) / kBlock);
  HIP_TRY(ev.run(st, kSpanFused, [&] {
    if (bounded_scene(s))
      hipLaunchKernelGGL((k_render_fused<true, true>), dim3(grid), dim3(kBlock), 0, st, a, s->ds, s->dl,
                         s->mats.ptr, s->node_info.ptr, s->n_materials, s->n_light_tris, b.pc,
                         b.shade_list, b.partial, b.cnt);
    else if (s->lds_geometry)
      hipLaunchKernelGGL((k_render_fused<true, false>), dim3(grid), dim3(kBlock), 0, st, a, s->ds, s->dl,
                         s->mats.ptr, s->node_info.ptr, s->n_materials, s->n_light_tris, b.pc,
                         b.shade_list, b.partial, b.cnt);
    else
      hipLaunchKernelGGL((k_render_fused<false, false>), dim3(grid), dim3(kBlock), 0, st, a, s->ds, s->dl,
                         s->mats.ptr, s->node_info.ptr, s->n_materials, s->n_light_tris, b.pc,
                         b.shade_list, b.partial, b.cnt);
  }));
  return RTR_OK;
}

int enqueue_resolve(const RenderPlan &pl, const RenderBuffers &b, float *d_rgb, hipStream_t st,
                    EventSpans &ev) {
  HIP_TRY(ev.run(st, kSpanOther, [&] {
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(pl.a.n_pix)), dim3(kBlock), 0, st, pl.a, b.partial, d_rgb);
  }));
  return RTR_OK;
}

// ---- device memory of the entry point that allocates nothing ----
// Largest render it takes, and how its scratch array is cut up (every part 256-byte
// aligned): counters, five per-pixel float4 arrays, the pixel list, the partial sums.
constexpr size_t kNoAllocMaxPixels = (size_t)1 << 18;  // e.g. 512 x 512
constexpr size_t kNoAllocMaxUnits = (size_t)1 << 20;   // pixels x partial sums per pixel
constexpr size_t kNoAllocCounters = 256;
constexpr size_t kNoAllocBytes = kNoAllocCounters + 5 * 16 * kNoAllocMaxPixels + 4 * kNoAllocMaxPixels +
                                 16 * kNoAllocMaxUnits;
static_assert(sizeof(Counters) <= kNoAllocCounters, "counters outgrew their part of the scratch array");

}  // namespace

// Host code must never use this name as an address: every device holds its own copy, and
// its address there comes from hipGetSymbolAddress (noalloc_scratch()).
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNoAllocBytes];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};

// The scratch array's address on `device`, which must be the current device.  Resolved
// once per device (scene creation does it, so that a render has nothing left to set up).
constexpr int kNoAllocDevices = 64;  // a power of two, at least the GPUs of a node
static std::atomic<int> g_noalloc_in_flight[kNoAllocDevices];
int noalloc_scratch(int device, unsigned char **out) {
  if (device < 0 || device >= kMaxDevices) return fail(RTR_ERR_UNSUPPORTED, "device index beyond 63");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void *p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch)));
    if (!p) return fail(RTR_ERR_DEVICE, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = (unsigned char *)p;
  }
  *out = g_scratch_ptr[device];
  return RTR_OK;
}

// The two streams of `device` (the current device) that every scene on it launches on.
// Made once per device and kept for the life of the process: the runtime deals streams to
// its few hardware queues (four per process by default) by what is in use at the moment a
// stream is made, so a pair made for a later scene can come to share one queue -- with each
// other or with a busy stream -- and the two halves of the wavefront loop then wait on each
// other's queue barriers (measured: the large-scene render 334 -> 414 ms with one idle scene
// made before it).  The first pair gets two queues of its own.  Work of different scenes on
// one device is ordered by these streams; a scene is still used from one thread at a time.
std::mutex g_stream_mutex;
hipStream_t g_streams[kMaxDevices][2] = {};
int device_streams(int device, hipStream_t *first, hipStream_t *second) {
  if (device < 0 || device >= kMaxDevices) return fail(RTR_ERR_UNSUPPORTED, "device index beyond 63");
  std::lock_guard<std::mutex> lock(g_stream_mutex);
  for (hipStream_t &st : g_streams[device])
    if (!st && hipStreamCreate(&st) != hipSuccess) {
      st = nullptr;
      return fail(RTR_ERR_DEVICE, "hipStreamCreate failed");
    }
  *first = g_streams[device][0];
  *second = g_streams[device][1];
  return RTR_OK;
}

RenderBuffers scratch_buffers(unsigned char *base) {
  RenderBuffers b;
  unsigned char *p = base;
  b.cnt = (Counters *)p; p += kNoAllocCounters;
  float4 *px[5];
  for (auto &q : px) { q = (float4 *)p; p += 16 * kNoAllocMaxPixels; }
  b.pix_dir = px[0]; b.pix_hit = px[1];
  b.pc = PixelCache{px[2], px[3], px[4]};
  b.shade_list = (uint32_t *)p; p += 4 * kNoAllocMaxPixels;
  b.partial = (float4 *)p;
  return b;
}

// pairs of devices whose peer access has been switched on (or found impossible)
std::mutex g_peer_mutex;
bool g_peer_tried[kMaxDevices][kMaxDevices] = {};
void enable_peer_access(int a, int b) {
  if (a == b || a < 0 || b < 0 || a >= kMaxDevices || b >= kMaxDevices) return;
  std::lock_guard<std::mutex> lock(g_peer_mutex);
  if (g_peer_tried[a][b]) return;
  g_peer_tried[a][b] = g_peer_tried[b][a] = true;
  // best effort: without it hipMemcpyPeerAsync still works, staged through the host
  int can = 0;
  for (int k = 0; k < 2; k++) {
    const int from = k ? b : a, to = k ? a : b;
    if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can && hipSetDevice(from) == hipSuccess)
      (void)hipDeviceEnablePeerAccess(to, 0);
  }
  (void)hipGetLastError();  // "already enabled" is not an error worth keeping
}

}  // namespace

extern "C" {

const char *rtr_last_error(void) { return g_error.c_str(); }

int rtr_device_count(int *n) {
  if (!n) return fail(RTR_ERR_INVALID, "n is null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(RTR_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *n = c;
  return RTR_OK;
}

int rtr_scene_create(const rtr_scene_desc *d, int device, rtr_scene **out) {
  if (!d || !out) return fail(RTR_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->n_nodes < 0 || d->n_prims < 0 || d->n_tris < 0 || d->n_spheres < 0 ||
      d->n_materials < 0 || d->n_lights < 0)
    return fail(RTR_ERR_INVALID, "negative count");
  if (d->n_tris >= (1 << 30) || d->n_spheres >= (1 << 30))
    return fail(RTR_ERR_INVALID, "too many primitives");
  // traversal packs a node index and a lane into 32 bits
  if (d->n_nodes > (1 << 26)) return fail(RTR_ERR_INVALID, "too many nodes (at most 2^26)");
  if (d->n_nodes > 0) {
    if (int rc = check_xyz(d->node_bmin, "node_bmin")) return rc;
    if (int rc = check_xyz(d->node_bmax, "node_bmax")) return rc;
    if (!d->node_left || !d->node_right || !d->node_prim)
      return fail(RTR_ERR_INVALID, "node links are null");
    if (d->root < 0 || d->root >= d->n_nodes) return fail(RTR_ERR_INVALID, "root out of range");
  }
  if (d->n_prims > 0 && (!d->prim_kind || !d->prim_index || !d->prim_material))
    return fail(RTR_ERR_INVALID, "prim arrays are null");
  if (d->n_tris > 0) {
    if (int rc = check_xyz(d->tri_p0, "tri_p0")) return rc;
    if (int rc = check_xyz(d->tri_p1, "tri_p1")) return rc;
    if (int rc = check_xyz(d->tri_p2, "tri_p2")) return rc;
    if (int rc = check_xyz(d->tri_n, "tri_n")) return rc;
    if (!d->tri_area) return fail(RTR_ERR_INVALID, "tri_area is null");
  }
  if (d->n_spheres > 0) {
    if (int rc = check_xyz(d->sph_c, "sph_c")) return rc;
    if (!d->sph_r || !d->sph_area) return fail(RTR_ERR_INVALID, "sphere arrays are null");
  }
  if (d->n_materials > 0) {
    if (int rc = check_xyz(d->mat_emission, "mat_emission")) return rc;
    if (int rc = check_xyz(d->mat_kd, "mat_kd")) return rc;
    if (int rc = check_xyz(d->mat_ks, "mat_ks")) return rc;
    if (!d->mat_kind || !d->mat_ior || !d->mat_roughness)
      return fail(RTR_ERR_INVALID, "material arrays are null");
  }
  if (d->n_lights > 0 && (!d->light_kind || !d->light_prim || !d->light_area ||
                          !d->light_tri_begin))
    return fail(RTR_ERR_INVALID, "light arrays are null");

  // ---- materials: a handful, laid out on the host ----
  std::vector<DeviceMaterial> mats(d->n_materials);
  for (int m = 0; m < d->n_materials; m++) {
    int kind = d->mat_kind[m];
    if (kind < 0 || kind > 2) return fail(RTR_ERR_INVALID, "unknown material kind");
    float ex = d->mat_emission[0][m], ey = d->mat_emission[1][m], ez = d->mat_emission[2][m];
    // Material::hasEmission, include/Material.h:65 (float against a double literal)
    bool emissive = (double)ex > 1e-4 || (double)ey > 1e-4 || (double)ez > 1e-4;
    uint32_t tag = (uint32_t)kind | (emissive ? 0x100u : 0u);
    float tagf;
    memcpy(&tagf, &tag, 4);
    mats[m].e_kind = make_float4(ex, ey, ez, tagf);
    mats[m].kd_ior = make_float4(d->mat_kd[0][m], d->mat_kd[1][m], d->mat_kd[2][m], d->mat_ior[m]);
    mats[m].ks_rough = make_float4(d->mat_ks[0][m], d->mat_ks[1][m], d->mat_ks[2][m], d->mat_roughness[m]);
  }
  // node_info packs prim id | material << 16 | emissive << 30 (small scenes, see hit_info)
  const bool info_fits = d->n_prims <= 0xffff && d->n_materials <= 0x3fff;

  // ---- lights: Scene::lights, a short list ----
  std::vector<float4> lights(d->n_lights), light_le(d->n_lights, make_float4(0, 0, 0, 0));
  std::vector<int32_t> light_tris;
  float total_area = 0;
  for (int l = 0; l < d->n_lights; l++) {
    float4 rec;
    rec.x = d->light_area[l];
    total_area += d->light_area[l];  // fp32 running sum, src/Scene.cpp:17-20
    int32_t kind = d->light_kind[l], z = 0, w = 0;
    int32_t le_prim = -1;  // the primitive whose material gives the light's emission
    if (kind == RTR_PRIM_SPHERE) {
      z = d->light_prim[l];
      if (z < 0 || z >= d->n_prims || d->prim_kind[z] != RTR_PRIM_SPHERE)
        return fail(RTR_ERR_INVALID, "sphere light does not name a sphere prim");
      le_prim = z;
      z = d->prim_index[z];  // the kernels want the sphere's row
      const float one_by_area = 1.0f / d->light_area[l];  // Sphere::sample's pdf, src/Sphere.cpp:47
      memcpy(&w, &one_by_area, 4);
    } else if (kind == RTR_PRIM_TRIANGLE) {
      int b = d->light_tri_begin[l], e = d->light_tri_begin[l + 1];
      if (b < 0 || e < b || (e > b && !d->light_tris))
        return fail(RTR_ERR_INVALID, "bad light_tri_begin");
      z = (int32_t)light_tris.size();
      for (int k = b; k < e; k++) {
        int p = d->light_tris[k];
        if (p < 0 || p >= d->n_prims || d->prim_kind[p] != RTR_PRIM_TRIANGLE)
          return fail(RTR_ERR_INVALID, "mesh light lists a non-triangle prim");
        light_tris.push_back(d->prim_index[p]);
        le_prim = p;  // samples always land on the last leaf (rtr_shade.hpp: sample_object)
      }
      w = (int32_t)light_tris.size();
    } else {
      return fail(RTR_ERR_INVALID, "unknown light_kind");
    }
    if (le_prim >= 0) {
      int m = d->prim_material[le_prim];
      if (m < 0 || m >= d->n_materials) return fail(RTR_ERR_INVALID, "prim_material out of range");
      light_le[l] = make_float4(d->mat_emission[0][m], d->mat_emission[1][m], d->mat_emission[2][m], 0);
    }
    memcpy(&rec.y, &kind, 4);
    memcpy(&rec.z, &z, 4);
    memcpy(&rec.w, &w, 4);
    lights[l] = rec;
  }

  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RTR_ERR_INVALID, "device index out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(RTR_ERR_DEVICE, "hipSetDevice failed");

  rtr_scene *s = new rtr_scene();
  s->device = device;
  auto bail = [&](int rc) {
    rtr_scene_destroy(s);
    return rc;
  };
  int rc;
  if ((rc = device_streams(device, &s->stream, &s->stream2))) return bail(rc);
  if ((rc = upload(s->mats, mats))) return bail(rc);
  if ((rc = upload(s->lights, lights))) return bail(rc);
  if ((rc = upload(s->light_le, light_le))) return bail(rc);
  if ((rc = upload(s->light_tris, light_tris))) return bail(rc);
  // everything sized by the tree or by the primitives is laid out on the device
  BuildFlags flags{};
  int32_t n_leaves = 0;
  if ((rc = build_scene_arrays(s, d, info_fits, flags, n_leaves))) return bail(rc);
  const bool fast_planes = flags.slow_planes == 0, nested_boxes = flags.not_nested == 0;
  s->n_materials = d->n_materials;
  s->ds.nodes = s->nodes.ptr;
  s->ds.tris = s->tris.ptr;
  s->ds.sphs = s->sphs.ptr;
  s->ds.tri_prim = s->tri_prim.ptr;
  s->ds.sph_prim = s->sph_prim.ptr;
  s->ds.prim_kind_row = s->prim_kind_row.ptr;
  s->ds.prim_material = s->prim_material.ptr;
  s->ds.n_nodes = d->n_nodes;
  s->ds.n_prims = d->n_prims;
  s->ds.fast_planes = fast_planes ? 1 : 0;
  s->ds.n_tris = d->n_tris;
  s->ds.n_sphs = d->n_spheres;
  s->ds.leaves = s->leaves.ptr;
  s->ds.parent = s->parent.ptr;
  s->ds.n_leaves = n_leaves;
  s->ds.nested_boxes = nested_boxes ? 1 : 0;
  s->ds.items = s->items.ptr;
  s->ds.pair_second = s->pair_second.ptr;
  s->ds.n_items = (int32_t)flags.n_items;
  s->ds.n_pairs = (int32_t)flags.n_pairs;
  s->ds.near_prims = flags.far_prims == 0 ? 1 : 0;
  s->lds_geometry = d->n_nodes <= kLdsNodes && d->n_tris <= kLdsTris && d->n_spheres <= kLdsSphs &&
                    info_fits && fits_lds_shade(d->n_materials, d->n_lights, (int)light_tris.size());
  s->n_light_tris = (int)light_tris.size();
  s->dl.light = s->lights.ptr;
  s->dl.light_le = s->light_le.ptr;
  s->dl.light_tris = s->light_tris.ptr;
  s->dl.tri_area = s->tri_area.ptr;
  s->dl.n_lights = d->n_lights;
  s->dl.total_area = total_area;
  // set up now what rtr_render_pt_device_noalloc must not do later: this device's address
  // of the scratch array and the fused kernel's launch shape
  if ((rc = noalloc_scratch(device, &s->noalloc_scratch))) return bail(rc);
  if ((rc = fused_grid(s))) return bail(rc);
  *out = s;
  return RTR_OK;
}

int rtr_scene_read_tree(rtr_scene *s, int32_t *n_nodes, int32_t *n_leaves, float *nodes8,
                        int32_t *parent, float *leaves16, int32_t *flags2) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n_nodes) *n_nodes = s->ds.n_nodes;
  if (n_leaves) *n_leaves = s->ds.n_leaves;
  if (flags2) {
    flags2[0] = s->ds.fast_planes;
    flags2[1] = s->ds.nested_boxes;
  }
  DeviceGuard guard(s->device);
  const size_t N = (size_t)s->ds.n_nodes, L = (size_t)s->ds.n_leaves;
  if (nodes8 && N) HIP_TRY(hipMemcpy(nodes8, s->nodes.ptr, N * 32, hipMemcpyDeviceToHost));
  if (parent && N) HIP_TRY(hipMemcpy(parent, s->parent.ptr, N * 4, hipMemcpyDeviceToHost));
  if (leaves16 && L) HIP_TRY(hipMemcpy(leaves16, s->leaves.ptr, L * 64, hipMemcpyDeviceToHost));
  return RTR_OK;
}

int rtr_scene_read_pairs(rtr_scene *s, int32_t *n_records, int32_t *n_pairs, float *records16,
                         float *seconds12) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n_records) *n_records = s->ds.n_items;
  if (n_pairs) *n_pairs = s->ds.n_pairs;
  DeviceGuard guard(s->device);
  const size_t R = (size_t)s->ds.n_items, P = (size_t)s->ds.n_pairs;
  if (records16 && R) HIP_TRY(hipMemcpy(records16, s->items.ptr, R * 64, hipMemcpyDeviceToHost));
  if (seconds12 && P) HIP_TRY(hipMemcpy(seconds12, s->pair_second.ptr, P * 48, hipMemcpyDeviceToHost));
  return RTR_OK;
}

void rtr_scene_destroy(rtr_scene *s) {
  if (!s) return;
  DeviceGuard guard(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);  // the device's streams outlive the scene
  if (s->stream2) (void)hipStreamSynchronize(s->stream2);
  s->nodes.release(); s->tris.release(); s->sphs.release(); s->lights.release();
  s->leaves.release(); s->items.release(); s->pair_second.release(); s->parent.release(); s->node_info.release(); s->light_le.release();
  s->tri_prim.release(); s->sph_prim.release(); s->prim_kind_row.release();
  s->prim_material.release(); s->light_tris.release(); s->tri_area.release(); s->mats.release();
  s->ws.release();
  s->trace_next.release();
  for (hipEvent_t e : s->trace_done)
    if (e) (void)hipEventDestroy(e);
  s->ph_counts.release(); s->ph_paths_used.release(); s->ph_sums.release(); s->ph_offs.release();
  s->ph_first.release(); s->ph_wave_count.release(); s->ph_stage.release();
  s->ph_total.release();
  for (auto &b : s->io_f) b.release();
  for (auto &b : s->io_i) b.release();
  delete s;
}

int rtr_trace_closest_device(rtr_scene *s, const float *d_org, const float *d_dir, size_t n,
                             int32_t *d_prim, float *d_t, float *d_pos, float *d_nrm,
                             void *stream) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n == 0) return RTR_OK;
  if (!d_org || !d_dir || !d_prim || !d_t) return fail(RTR_ERR_INVALID, "null ray or result array");
  if (n > (size_t)0x7fffffff * kBlock) return fail(RTR_ERR_INVALID, "batch too large");
  DeviceGuard guard(s->device);
  if (s->lds_geometry) {
    hipLaunchKernelGGL(k_trace_api<true>, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       s->ds, d_org, d_dir, n, d_prim, d_t, d_pos, d_nrm);
  } else {
    unsigned grid = 0, turn = 0;
    unsigned long long *next = nullptr;
    int rc = stream_launch(s, k_trace_api_stream, s->stream_grid, n, (hipStream_t)stream, grid, next, turn);
    if (rc) return rc;
    hipLaunchKernelGGL(k_trace_api_stream, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, s->ds,
                       d_org, d_dir, (unsigned long long)n, d_prim, d_t, d_pos, d_nrm, next);
    HIP_TRY(hipGetLastError());
    return stream_launched(s, turn, (hipStream_t)stream);
  }
  HIP_TRY(hipGetLastError());
  return RTR_OK;
}

int rtr_self_test(int device, uint64_t *mismatches) {
  if (!mismatches) return fail(RTR_ERR_INVALID, "mismatches is null");
  *mismatches = 0;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RTR_ERR_INVALID, "device index out of range");
  DeviceGuard guard(device);
  unsigned long long *d_bad = nullptr;
  HIP_TRY(hipMalloc((void **)&d_bad, sizeof(*d_bad)));
  hipError_t e = hipMemset(d_bad, 0, sizeof(*d_bad));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_selftest_reciprocal, dim3(65536), dim3(kBlock), 0, nullptr, d_bad);
    hipLaunchKernelGGL(k_selftest_div_by_pi, dim3(65536), dim3(kBlock), 0, nullptr, d_bad);
    // every float in [2^-100, 2^100], a quarter of them per launch (the last one takes 2^100 itself)
    const uint32_t lo = (127u - 100u) << 23, total = 200u << 23;
    for (uint32_t done = 0; done < total; done += total / 4) {
      const uint32_t count = total / 4 + (done + total / 4 == total ? 1u : 0u);
      hipLaunchKernelGGL(k_selftest_sqrt, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, lo + done,
                         count, d_bad);
    }
    hipLaunchKernelGGL(k_selftest_unit_near, dim3((1u << 20) / kBlock), dim3(kBlock), 0, nullptr, kNearOneLo,
                       kNearOneHi, d_bad);
    for (uint32_t round = 0; round < 16; round++)  // 2^32 vectors
      hipLaunchKernelGGL(k_selftest_unit, dim3(1u << 20), dim3(kBlock), 0, nullptr, round * 7919u + 1u, d_bad);
    for (uint32_t round = 0; round < 4; round++)  // 2^30 colours
      hipLaunchKernelGGL(k_selftest_div_div, dim3(1u << 20), dim3(kBlock), 0, nullptr, round * 104729u + 3u, d_bad);
    e = hipGetLastError();
  }
  unsigned long long bad = 0;
  if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
  (void)hipFree(d_bad);
  HIP_TRY(e);
  *mismatches = bad;
  return RTR_OK;
}

int rtr_trace_count_work(rtr_scene *s, const float *d_org, const float *d_dir, size_t n,
                         uint64_t *boxes, uint64_t *triangles, uint64_t *spheres) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (boxes) *boxes = 0;
  if (triangles) *triangles = 0;
  if (spheres) *spheres = 0;
  if (n == 0) return RTR_OK;
  if (!d_org || !d_dir) return fail(RTR_ERR_INVALID, "null ray array");
  DeviceGuard guard(s->device);
  HIP_TRY(s->ph_sums.reserve(3));  // three 64-bit sums
  HIP_TRY(hipMemsetAsync(s->ph_sums.ptr, 0, 3 * sizeof(unsigned long long), s->stream));
  hipLaunchKernelGGL(k_count_work, dim3(blocks_for(n)), dim3(kBlock), 0, s->stream, s->ds, d_org, d_dir,
                     n, s->ph_sums.ptr);
  HIP_TRY(hipGetLastError());
  unsigned long long out[3] = {0, 0, 0};
  HIP_TRY(hipMemcpyAsync(out, s->ph_sums.ptr, sizeof(out), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (boxes) *boxes = out[0];
  if (triangles) *triangles = out[1];
  if (spheres) *spheres = out[2];
  return RTR_OK;
}

int rtr_trace_closest(rtr_scene *s, const float *org, const float *dir, size_t n, int32_t *prim,
                      float *t, float *pos, float *nrm) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n == 0) return RTR_OK;
  if (!org || !dir || !prim || !t) return fail(RTR_ERR_INVALID, "null ray or result array");
  DeviceGuard guard(s->device);
  HIP_TRY(s->io_f[0].reserve(3 * n));
  HIP_TRY(s->io_f[1].reserve(3 * n));
  HIP_TRY(s->io_f[2].reserve(n));
  HIP_TRY(s->io_i[0].reserve(n));
  if (pos) HIP_TRY(s->io_f[3].reserve(3 * n));
  if (nrm) HIP_TRY(s->io_f[4].reserve(3 * n));
  HIP_TRY(hipMemcpyAsync(s->io_f[0].ptr, org, 3 * n * sizeof(float), hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[1].ptr, dir, 3 * n * sizeof(float), hipMemcpyHostToDevice, s->stream));
  int rc = rtr_trace_closest_device(s, s->io_f[0].ptr, s->io_f[1].ptr, n, s->io_i[0].ptr,
                                    s->io_f[2].ptr, pos ? s->io_f[3].ptr : nullptr,
                                    nrm ? s->io_f[4].ptr : nullptr, s->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(prim, s->io_i[0].ptr, n * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(t, s->io_f[2].ptr, n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  if (pos) HIP_TRY(hipMemcpyAsync(pos, s->io_f[3].ptr, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  if (nrm) HIP_TRY(hipMemcpyAsync(nrm, s->io_f[4].ptr, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_material_eval(rtr_scene *s, const int32_t *material, const float *wi, const float *wo,
                      const float *nn, size_t count, float *f, float *pdf) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (count == 0) return RTR_OK;
  if (!material || !wi || !wo || !nn) return fail(RTR_ERR_INVALID, "null input array");
  for (size_t i = 0; i < count; i++)
    if (material[i] < 0 || material[i] >= s->n_materials)
      return fail(RTR_ERR_INVALID, "material id out of range");
  DeviceGuard guard(s->device);
  HIP_TRY(s->io_i[0].reserve(count));
  for (int k = 0; k < 4; k++) HIP_TRY(s->io_f[k].reserve(3 * count));
  HIP_TRY(s->io_f[4].reserve(count));
  HIP_TRY(hipMemcpyAsync(s->io_i[0].ptr, material, count * 4, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[0].ptr, wi, 12 * count, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[1].ptr, wo, 12 * count, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[2].ptr, nn, 12 * count, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_material_eval, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream,
                     s->mats.ptr, s->io_i[0].ptr, s->io_f[0].ptr, s->io_f[1].ptr, s->io_f[2].ptr,
                     count, s->io_f[3].ptr, s->io_f[4].ptr);
  HIP_TRY(hipGetLastError());
  if (f) HIP_TRY(hipMemcpyAsync(f, s->io_f[3].ptr, 12 * count, hipMemcpyDeviceToHost, s->stream));
  if (pdf) HIP_TRY(hipMemcpyAsync(pdf, s->io_f[4].ptr, 4 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_material_sample(rtr_scene *s, const int32_t *material, const float *wi, const float *nn,
                        const float *u2, size_t count, float *wo) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (count == 0) return RTR_OK;
  if (!material || !wi || !nn || !u2 || !wo) return fail(RTR_ERR_INVALID, "null array");
  for (size_t i = 0; i < count; i++)
    if (material[i] < 0 || material[i] >= s->n_materials)
      return fail(RTR_ERR_INVALID, "material id out of range");
  DeviceGuard guard(s->device);
  HIP_TRY(s->io_i[0].reserve(count));
  for (int k = 0; k < 4; k++) HIP_TRY(s->io_f[k].reserve(3 * count));
  HIP_TRY(hipMemcpyAsync(s->io_i[0].ptr, material, count * 4, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[0].ptr, wi, 12 * count, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[1].ptr, nn, 12 * count, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[2].ptr, u2, 8 * count, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_material_sample, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream,
                     s->mats.ptr, s->io_i[0].ptr, s->io_f[0].ptr, s->io_f[1].ptr, s->io_f[2].ptr,
                     count, s->io_f[3].ptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(wo, s->io_f[3].ptr, 12 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_sample_light(rtr_scene *s, const float *u, int32_t stride, size_t count, float *pos,
                     float *nrm, float *pdf, int32_t *prim, int32_t *used) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (count == 0) return RTR_OK;
  if (!u || stride <= 0 || !pos || !nrm || !pdf || !prim || !used)
    return fail(RTR_ERR_INVALID, "null array or bad stride");
  DeviceGuard guard(s->device);
  HIP_TRY(s->io_f[0].reserve((size_t)stride * count));
  HIP_TRY(s->io_f[1].reserve(3 * count));
  HIP_TRY(s->io_f[2].reserve(3 * count));
  HIP_TRY(s->io_f[3].reserve(count));
  HIP_TRY(s->io_i[0].reserve(count));
  HIP_TRY(s->io_i[1].reserve(count));
  HIP_TRY(hipMemcpyAsync(s->io_f[0].ptr, u, 4 * (size_t)stride * count, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_sample_light, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream, s->ds,
                     s->dl, s->io_f[0].ptr, (int)stride, count, s->io_f[1].ptr, s->io_f[2].ptr,
                     s->io_f[3].ptr, s->io_i[0].ptr, s->io_i[1].ptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pos, s->io_f[1].ptr, 12 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(nrm, s->io_f[2].ptr, 12 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(pdf, s->io_f[3].ptr, 4 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(prim, s->io_i[0].ptr, 4 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(used, s->io_i[1].ptr, 4 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_stream_uniform(rtr_scene *s, uint32_t seed, const uint32_t *pixel, const uint32_t *sample,
                       const uint32_t *bounce, const uint32_t *dim, size_t count, float *u) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (count == 0) return RTR_OK;
  if (!pixel || !sample || !bounce || !dim || !u) return fail(RTR_ERR_INVALID, "null array");
  DeviceGuard guard(s->device);
  for (int k = 0; k < 5; k++) HIP_TRY(s->io_f[k].reserve(count));
  const uint32_t *src[4] = {pixel, sample, bounce, dim};
  for (int k = 0; k < 4; k++)
    HIP_TRY(hipMemcpyAsync(s->io_f[k].ptr, src[k], 4 * count, hipMemcpyHostToDevice, s->stream));
  hipLaunchKernelGGL(k_stream_uniform, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream, seed,
                     (const uint32_t *)s->io_f[0].ptr, (const uint32_t *)s->io_f[1].ptr,
                     (const uint32_t *)s->io_f[2].ptr, (const uint32_t *)s->io_f[3].ptr, count,
                     s->io_f[4].ptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(u, s->io_f[4].ptr, 4 * count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_check_connectable(rtr_scene *s, const float *a_xyz, const float *b_xyz, size_t count,
                          uint8_t *connected) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (count == 0) return RTR_OK;
  if (!a_xyz || !b_xyz || !connected) return fail(RTR_ERR_INVALID, "null array");
  DeviceGuard guard(s->device);
  HIP_TRY(s->io_f[0].reserve(3 * count));
  HIP_TRY(s->io_f[1].reserve(3 * count));
  HIP_TRY(s->io_i[0].reserve((count + 3) / 4));
  HIP_TRY(hipMemcpyAsync(s->io_f[0].ptr, a_xyz, 12 * count, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->io_f[1].ptr, b_xyz, 12 * count, hipMemcpyHostToDevice, s->stream));
  unsigned char *out = (unsigned char *)s->io_i[0].ptr;
  if (s->lds_geometry) {
    hipLaunchKernelGGL(k_check_connectable<true>, dim3(blocks_for(count)), dim3(kBlock), 0, s->stream,
                       s->ds, (const float *)s->io_f[0].ptr, (const float *)s->io_f[1].ptr, count, out);
  } else {
    unsigned grid = 0, turn = 0;
    unsigned long long *next = nullptr;
    int rc = stream_launch(s, k_check_connectable_stream, s->connect_grid, count, s->stream, grid, next, turn);
    if (rc) return rc;
    hipLaunchKernelGGL(k_check_connectable_stream, dim3(grid), dim3(kBlock), 0, s->stream, s->ds,
                       (const float *)s->io_f[0].ptr, (const float *)s->io_f[1].ptr,
                       (unsigned long long)count, out, next);
    HIP_TRY(hipGetLastError());
    if ((rc = stream_launched(s, turn, s->stream))) return rc;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(connected, out, count, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return RTR_OK;
}

int rtr_create_photon_map_device(rtr_scene *s, uint64_t photon_num, uint32_t seed, float *d_pos,
                                 float *d_wi, float *d_power, uint64_t *n_stored,
                                 uint32_t *paths_used) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n_stored) *n_stored = 0;
  if (paths_used) *paths_used = 0;
  if (photon_num == 0) return RTR_OK;
  if (!d_pos || !d_wi || !d_power) return fail(RTR_ERR_INVALID, "null photon array");
  DeviceGuard guard(s->device);
  if (s->dl.n_lights == 0) return RTR_OK;  // the reference leaves the map empty (:413-415)
  hipStream_t st = s->stream;
  HIP_TRY(s->ph_total.reserve(1));
  HIP_TRY(s->ph_paths_used.reserve(1));
  HIP_TRY(hipMemsetAsync(s->ph_paths_used.ptr, 0, sizeof(uint32_t), st));
  uint64_t filled = 0, paths = 0;
  int barren = 0;
  while (filled < photon_num) {
    // batch: enough paths for what is missing at the yield seen so far (first guess: two
    // photons per path), at least 4 Ki, at most 4 Mi
    const double yield = paths ? std::max((double)filled / (double)paths, 0.05) : 2.0;
    uint64_t want = (uint64_t)((double)(photon_num - filled) / yield * 1.05) + 1024;
    want = std::min<uint64_t>(std::max<uint64_t>(want, 4096), (uint64_t)1 << 22);
    const uint32_t n_paths = (uint32_t)((want + kBlock - 1) / kBlock * kBlock);
    if (paths + n_paths > 0xffffffffull) return fail(RTR_ERR_INVALID, "photon map needs too many emission paths");
    const unsigned nb = n_paths / kBlock;
    HIP_TRY(s->ph_counts.reserve(n_paths));
    HIP_TRY(s->ph_sums.reserve(nb));
    HIP_TRY(s->ph_offs.reserve(nb));
    PhotonArgs a{seed, (uint32_t)paths, n_paths, filled, photon_num};
    auto paths_pass = [&](int pass, PhotonStage stage) {
      auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(nb), dim3(kBlock), 0, st, a, s->ds, s->dl, s->mats.ptr, s->ph_counts.ptr,
                           s->ph_sums.ptr, s->ph_offs.ptr, d_pos, d_wi, d_power, s->ph_paths_used.ptr, stage);
      };
      if (s->lds_geometry) {
        if (pass == kPhotonCount) launch(k_photon_paths<true, kPhotonCount>);
        else if (pass == kPhotonWrite) launch(k_photon_paths<true, kPhotonWrite>);
        else launch(k_photon_paths<true, kPhotonStage>);
      } else {
        if (pass == kPhotonCount) launch(k_photon_paths<false, kPhotonCount>);
        else if (pass == kPhotonWrite) launch(k_photon_paths<false, kPhotonWrite>);
        else launch(k_photon_paths<false, kPhotonStage>);
      }
    };
    // One pass: the paths run once and stage their photons (every wave in room of its own), the
    // counts are scanned, the photons move to their places.  A batch in which some wave ran out
    // of room is run again the slow way (count, scan, run again and write).
    const size_t n_waves = n_paths / 64;
    // (RTR_PHOTON_WAVE_ROOM: a test's way to a batch that runs out of room)
    static const unsigned room = [] {
      const char *e = getenv("RTR_PHOTON_WAVE_ROOM");
      const long v = e ? atol(e) : 0;
      return v > 0 && v < (long)kPhotonWaveRoom ? (unsigned)v : kPhotonWaveRoom;
    }();
    HIP_TRY(s->ph_stage.reserve(3 * n_waves * room));
    HIP_TRY(s->ph_first.reserve(n_paths));
    HIP_TRY(s->ph_wave_count.reserve(n_waves + 1));
    uint32_t *overflow = s->ph_wave_count.ptr + n_waves;
    HIP_TRY(hipMemsetAsync(overflow, 0, sizeof(uint32_t), st));
    PhotonStage stage{room, s->ph_stage.ptr, s->ph_wave_count.ptr, overflow};
    paths_pass(kPhotonStage, stage);
    hipLaunchKernelGGL(k_photon_scan, dim3(1), dim3(kBlock), 0, st, s->ph_sums.ptr, s->ph_offs.ptr, nb,
                       s->ph_total.ptr);
    hipLaunchKernelGGL(k_photon_offsets, dim3(nb), dim3(kBlock), 0, st, a, s->ph_counts.ptr, s->ph_offs.ptr,
                       s->ph_first.ptr, s->ph_paths_used.ptr);
    hipLaunchKernelGGL(k_photon_scatter, dim3((unsigned)((n_waves * room + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, st, a, stage, s->ph_first.ptr, d_pos, d_wi, d_power);
    uint32_t overflowed = 0;
    HIP_TRY(hipMemcpyAsync(&overflowed, overflow, sizeof(overflowed), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (overflowed) {
      paths_pass(kPhotonCount, stage);
      hipLaunchKernelGGL(k_photon_scan, dim3(1), dim3(kBlock), 0, st, s->ph_sums.ptr, s->ph_offs.ptr, nb,
                         s->ph_total.ptr);
      paths_pass(kPhotonWrite, stage);
    }
    HIP_TRY(hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, s->ph_total.ptr, sizeof(total), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    filled += total;
    paths += n_paths;
    barren = total == 0 ? barren + 1 : 0;
    // lights exist but no path ever stores a photon: the reference would loop for ever
    if (barren >= 4) return fail(RTR_ERR_INVALID, "no emission path stores a photon in this scene");
  }
  uint32_t used = 0;
  HIP_TRY(hipMemcpy(&used, s->ph_paths_used.ptr, sizeof(used), hipMemcpyDeviceToHost));
  if (n_stored) *n_stored = std::min<uint64_t>(filled, photon_num);
  if (paths_used) *paths_used = used;
  return RTR_OK;
}

int rtr_create_photon_map(rtr_scene *s, uint64_t photon_num, uint32_t seed, float *pos_xyz,
                          float *wi_xyz, float *power_xyz, uint64_t *n_stored, uint32_t *paths_used) {
  if (!s) return fail(RTR_ERR_INVALID, "scene is null");
  if (n_stored) *n_stored = 0;
  if (paths_used) *paths_used = 0;
  if (photon_num == 0) return RTR_OK;
  if (!pos_xyz || !wi_xyz || !power_xyz) return fail(RTR_ERR_INVALID, "null photon array");
  DeviceGuard guard(s->device);
  for (int k = 0; k < 3; k++) HIP_TRY(s->io_f[k].reserve(3 * photon_num));
  uint64_t stored = 0;
  int rc = rtr_create_photon_map_device(s, photon_num, seed, s->io_f[0].ptr, s->io_f[1].ptr,
                                        s->io_f[2].ptr, &stored, paths_used);
  if (rc) return rc;
  float *dst[3] = {pos_xyz, wi_xyz, power_xyz};
  for (int k = 0; k < 3; k++)
    if (stored) HIP_TRY(hipMemcpy(dst[k], s->io_f[k].ptr, 12 * stored, hipMemcpyDeviceToHost));
  if (n_stored) *n_stored = stored;
  return RTR_OK;
}

int rtr_partition_columns(int32_t width, int32_t part_index, int32_t part_count,
                          int32_t tile_cols, int32_t *n_columns, int32_t *columns) {
  if (width < 0 || (part_count > 1 && (part_index < 0 || part_index >= part_count)))
    return fail(RTR_ERR_INVALID, "bad partition");
  std::vector<int32_t> cols;
  partition_columns(width, part_index, part_count, tile_cols, cols);
  // the renderer never builds this list: its kernels map local columns to image columns
  // by formula (ColumnMap), and the count comes from count_columns()
  if ((int)cols.size() != count_columns(width, part_index, part_count, tile_cols > 0 ? tile_cols : 8))
    return fail(RTR_ERR_DEVICE, "internal: column count and column list disagree");
  if (n_columns) *n_columns = (int32_t)cols.size();
  if (columns) memcpy(columns, cols.data(), cols.size() * sizeof(int32_t));
  return RTR_OK;
}

int rtr_render_pt_device(rtr_scene *s, const rtr_camera *cam, const rtr_render_params *prm,
                         float *d_rgb, rtr_stats *stats) {
  if (!s || !cam || !prm || !d_rgb) return fail(RTR_ERR_INVALID, "null argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  auto t_start = std::chrono::steady_clock::now();
  RenderPlan pl;
  // unit budget: 2^27 partial sums (2 GiB); 128 per pixel up to about 1000 x 1000 pixels
  if (int rc = plan_render(s, cam, prm, (size_t)1 << 27, pl)) return rc;
  RenderArgs &a = pl.a;
  if (a.n_pix == 0) return RTR_OK;
  DeviceGuard guard(s->device);
  Workspace &ws = s->ws;
  hipStream_t st = s->stream;
  const size_t max_units = pl.units;
  const bool fused = pl.fused;
  // upper bound of the pool; the exact size follows from the number of pixels
  // that need sampling (known after k_classify)
  // (measured: 2^21 slots is best where the scene sits in LDS; where traversal reads it from
  // memory, 2^22 -- its launches then carry enough rays to amortise their tails -- and 2^23
  // once the render is long enough for the larger launches to outweigh the longer drain at
  // its end: 784 x 784 on a 1 M-triangle scene, 2^22 : 2^23 slots, 97 : 99 ms at 64 spp,
  // 337 : 324 ms at 256 spp, 1297 : 1225 ms at 1024 spp)
  const bool long_render = (uint64_t)a.n_pix * (uint64_t)a.spp >= ((uint64_t)1 << 27);
  size_t pool_target = prm->pool_size > 0 ? (size_t)prm->pool_size
                                          : ((size_t)1 << (s->lds_geometry ? 21 : long_render ? 23 : 22));
  size_t pool = pool_target;
  if (pool > max_units) pool = max_units;
  pool = (pool + (kLogicBlock * kGroup) - 1) / (kLogicBlock * kGroup) * (kLogicBlock * kGroup);
  if (pool > 0x7fffffffu) return fail(RTR_ERR_INVALID, "pool too large");

  HIP_TRY(ws.pix_dir.reserve(a.n_pix));
  HIP_TRY(ws.pix_hit.reserve(a.n_pix));
  HIP_TRY(ws.pc_pos.reserve(a.n_pix));
  HIP_TRY(ws.pc_nrm.reserve(a.n_pix));
  HIP_TRY(ws.pc_wi.reserve(a.n_pix));
  HIP_TRY(ws.shade_list.reserve(a.n_pix));
  HIP_TRY(ws.partial.reserve(max_units));
  HIP_TRY(ws.counters.reserve(1));
  if (!ws.host_counters) HIP_TRY(hipHostMalloc((void **)&ws.host_counters, sizeof(Counters)));
  size_t n_blocks = 0;
  if (!fused) {
    for (auto &b : ws.pool) HIP_TRY(b.reserve(pool));
    HIP_TRY(ws.queue.reserve(2 * pool));  // one segment of kSegment entries per k_logic workgroup
    HIP_TRY(ws.wave_res.reserve(pool / 64));
    n_blocks = pool / kLogicBlock;  // capacity; the launch uses what the schedule needs
    HIP_TRY(ws.block_counters.reserve(n_blocks));
    if (ws.host_block_cap < n_blocks) {
      if (ws.host_block_counters) (void)hipHostFree(ws.host_block_counters);
      ws.host_block_counters = nullptr;
      ws.host_block_cap = 0;
      HIP_TRY(hipHostMalloc((void **)&ws.host_block_counters, n_blocks * sizeof(BlockCounters)));
      ws.host_block_cap = n_blocks;
    }
  } else {
    if (int rc = fused_grid(s)) return rc;
  }

  RenderBuffers rb{ws.pix_dir.ptr, ws.pix_hit.ptr, ws.partial.ptr,
                   PixelCache{ws.pc_pos.ptr, ws.pc_nrm.ptr, ws.pc_wi.ptr}, ws.shade_list.ptr,
                   ws.counters.ptr};
  const PixelCache pc = rb.pc;
  PathPool pp{ws.pool[0].ptr, ws.pool[1].ptr, ws.pool[2].ptr, ws.pool[3].ptr, ws.pool[4].ptr,
              ws.pool[5].ptr, ws.pool[6].ptr, ws.pool[7].ptr, ws.pool[8].ptr, ws.pool[9].ptr,
              ws.pool[10].ptr, ws.pool[11].ptr, ws.pool[12].ptr, ws.pool[13].ptr};
  Counters *cnt = rb.cnt;
  BlockCounters *bc = ws.block_counters.ptr;

  // profile=1: every kernel is bracketed by a pair of events on the launch stream;
  // elapsed times are read back only after a synchronisation the render needs
  // anyway, so profiling adds event records but no extra waits.
  EventSpans ev;
  ev.on = prm->profile != 0;

  uint64_t trace_launches = 1, fused_launches = 0;  // 1: the primary rays
  if (int rc = enqueue_front(s, pl, rb, st, ev)) return rc;
  // The wavefront loop sizes its pool and its unit schedule from the number of
  // pixels that need sampling; the fused kernel reads that counter on the device, so
  // nothing waits between classification and sampling.
  uint64_t n_units = max_units;  // fused mode: an upper bound
  if (!fused) {
    HIP_TRY(hipMemcpyAsync(ws.host_counters, cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ev.harvest();
    n_units = (uint64_t)ws.host_counters->n_shade * a.n_chunks;
    a.n_shade = ws.host_counters->n_shade;
    a.n_units = (uint32_t)n_units;
  }
  if (n_units > 0 && !fused) {
    // no more slots than units (units are dealt on demand, see k_logic)
    size_t used = std::min<size_t>(pool_target, (size_t)n_units);
    used = (used + (kLogicBlock * kGroup) - 1) / (kLogicBlock * kGroup) * (kLogicBlock * kGroup);
    if (used > pool) return fail(RTR_ERR_DEVICE, "internal: pool smaller than planned");
    pool = used;
    n_blocks = pool / kLogicBlock;
  }
  a.pool = (int)pool;

  uint64_t iters = 0, total_rays = 0, total_vertices = 0;
  if (n_units > 0 && fused) {
    if (int rc = enqueue_fused(s, pl, rb, st, ev)) return rc;
    fused_launches = 1;
    iters = 1;
  } else if (n_units > 0) {
    const int batch = 16;
    // The pool is driven as two halves on two streams (streams = 2).  Path logic is
    // bound by HBM latency and traversal by fp32 issue, so one half's traversal
    // overlaps the other half's logic instead of each waiting for the other.
    const size_t n_groups = n_blocks / kGroup;
    const int n_halves = (n_groups >= 2 && prm->streams != 1) ? 2 : 1;
    const size_t grp_begin[3] = {0, n_halves == 2 ? n_groups / 2 : n_groups, n_groups};
    hipStream_t streams[2] = {st, s->stream2};
    unsigned queue_grid = 0;
    if (!s->lds_geometry) {
      if (int rc = resident_grid(s, k_trace_queue_stream, s->queue_grid)) return rc;
      // no more waves than there could be 64-ray batches in a launch
      const size_t most = (grp_begin[1] * kGroup * kSegment + kBlock - 1) / kBlock;
      queue_grid = (unsigned)std::min<size_t>((size_t)s->queue_grid, std::max<size_t>(most, 1));
      // with two streams, seven eighths of that: a resident grid that fills the device keeps
      // the other half's path logic waiting until its waves run out of rays; with some room
      // left the logic kernel runs beside it (1 M-triangle scene, 784 x 784: 97 -> 94 ms at
      // 64 spp, 324 -> 318 ms at 256 spp)
      if (n_halves == 2) queue_grid = std::max(1u, queue_grid - queue_grid / 8u);
    }
    for (;;) {
      for (int b = 0; b < batch; b++) {
        int parity = (int)(iters & 1);
        a.first = iters == 0 ? 1 : 0;
        for (int h = 0; h < n_halves; h++) {
          const unsigned g0 = (unsigned)grp_begin[h], ng = (unsigned)(grp_begin[h + 1] - grp_begin[h]);
          HIP_TRY(ev.run(streams[h], kSpanLogic, [&] {
            hipLaunchKernelGGL(k_logic, dim3(ng * kGroup), dim3(kLogicBlock), 0, streams[h], a, s->ds,
                               s->dl, s->mats.ptr, pc, pp, ws.shade_list.ptr, ws.queue.ptr, bc,
                               ws.partial.ptr, cnt, ws.wave_res.ptr, parity, (int)(g0 * kGroup));
          }));
          HIP_TRY(ev.run(streams[h], kSpanTrace, [&] {
            if (s->lds_geometry)
              hipLaunchKernelGGL(k_trace_queue<true>, dim3(ng * kGroupParts), dim3(kBlock), 0,
                                 streams[h], s->ds, pp, ws.queue.ptr, bc, parity, g0);
            else
              hipLaunchKernelGGL(k_trace_queue_stream, dim3(queue_grid), dim3(kBlock), 0, streams[h],
                                 s->ds, pp, ws.queue.ptr, bc, cnt, parity, g0 * kGroup, ng * kGroup);
          }));
          trace_launches++;
        }
        iters++;
      }
      if (n_halves == 2) HIP_TRY(hipStreamSynchronize(s->stream2));
      HIP_TRY(hipMemcpyAsync(ws.host_block_counters, bc, n_blocks * sizeof(BlockCounters),
                             hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      ev.harvest();
      uint64_t done = 0;
      total_rays = total_vertices = 0;
      for (size_t k = 0; k < n_blocks; k++) {
        done += ws.host_block_counters[k].done;
        total_rays += ws.host_block_counters[k].rays[0] + ws.host_block_counters[k].rays[1];
        total_vertices += ws.host_block_counters[k].vertices;
      }
      if (done >= n_units) break;
      if (iters > (uint64_t)1 << 40) return fail(RTR_ERR_DEVICE, "render did not terminate");
    }
  }
  if (int rc = enqueue_resolve(pl, rb, d_rgb, st, ev)) return rc;
  HIP_TRY(hipMemcpyAsync(ws.host_counters, cnt, sizeof(Counters), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  ev.harvest();
  if (fused) {
    total_rays = ws.host_counters->rays;
    total_vertices = ws.host_counters->vertices;
  }
  if (stats) {
    stats->samples = (uint64_t)a.n_pix * a.spp;
    stats->rays = total_rays + (uint64_t)a.n_pix;  // queue rays + primary rays
    stats->vertices = total_vertices;
    stats->iterations = iters;
    stats->chunk_spp = a.chunk_spp;
    stats->pool_size = fused ? 0 : a.pool;
    stats->mode = fused ? RTR_MODE_FUSED : RTR_MODE_WAVEFRONT;
    stats->n_columns = a.n_cols;
    stats->ms_trace = ev.ms[kSpanTrace];
    stats->ms_logic = ev.ms[kSpanLogic];
    stats->ms_other = ev.ms[kSpanOther];
    stats->trace_launches = trace_launches;
    stats->ms_fused = ev.ms[kSpanFused];
    stats->fused_launches = fused_launches;
    stats->n_devices = 1;
    stats->boxes_tested = ws.host_counters->boxes;
    stats->prims_tested = ws.host_counters->prims;
    stats->ms_total =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    stats->ms_render_min = stats->ms_render_max = stats->ms_total;
  }
  return RTR_OK;
}

int rtr_render_pt(rtr_scene *s, const rtr_camera *cam, const rtr_render_params *prm, float *rgb,
                  rtr_stats *stats) {
  if (!s || !cam || !prm || !rgb) return fail(RTR_ERR_INVALID, "null argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  auto t_start = std::chrono::steady_clock::now();
  RenderPlan pl;
  if (int rc = plan_render(s, cam, prm, (size_t)1 << 27, pl)) return rc;
  const size_t n = (size_t)pl.a.n_pix * 3;
  if (n == 0) return RTR_OK;
  DeviceGuard guard(s->device);
  HIP_TRY(s->ws.rgb.reserve(n));
  int rc = rtr_render_pt_device(s, cam, prm, s->ws.rgb.ptr, stats);
  if (rc) return rc;
  if (s->ws.reserve_host_rgb(n) == hipSuccess) {
    HIP_TRY(hipMemcpy(s->ws.host_rgb, s->ws.rgb.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
    memcpy(rgb, s->ws.host_rgb, n * sizeof(float));
  } else {
    (void)hipGetLastError();  // no pinned memory to be had: the slower direct copy
    HIP_TRY(hipMemcpy(rgb, s->ws.rgb.ptr, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (stats) {
    stats->ms_total =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    stats->ms_render_min = stats->ms_render_max = stats->ms_total;
  }
  return RTR_OK;
}

int rtr_render_pt_device_noalloc(rtr_scene *s, const rtr_camera *cam, const rtr_render_params *prm,
                                 float *d_rgb, rtr_stats *stats) {
  if (!s || !cam || !prm || !d_rgb) return fail(RTR_ERR_INVALID, "null argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  auto t_start = std::chrono::steady_clock::now();
  if (prm->profile) return fail(RTR_ERR_UNSUPPORTED, "profile=1 creates events: not in the call that allocates nothing");
  RenderPlan pl;
  if (int rc = plan_render(s, cam, prm, kNoAllocMaxUnits, pl)) return rc;
  if (!pl.fused)
    return fail(RTR_ERR_UNSUPPORTED, "the wavefront loop needs a path pool: only the fused kernel runs without allocating");
  if ((size_t)pl.a.n_pix > kNoAllocMaxPixels)
    return fail(RTR_ERR_INVALID, "more than 262144 pixels (2^18) in one call: beyond the fixed scratch array");
  if (pl.units > kNoAllocMaxUnits)
    return fail(RTR_ERR_INVALID, "pixels x partial sums per pixel beyond 1048576 (2^20): raise chunk_spp");
  if (pl.a.n_pix == 0) return RTR_OK;
  if (!s->noalloc_scratch || s->fused_grid == 0)
    return fail(RTR_ERR_DEVICE, "scene was created without the scratch array's address");
  // one call at a time per GPU: a second one would write over the first one's scratch array
  struct InFlight {
    std::atomic<int> &flag;
    bool mine;
    explicit InFlight(std::atomic<int> &f) : flag(f), mine(f.exchange(1) == 0) {}
    ~InFlight() { if (mine) flag.store(0); }
  } in_flight(g_noalloc_in_flight[s->device & (kNoAllocDevices - 1)]);
  if (!in_flight.mine)
    return fail(RTR_ERR_INVALID, "another rtr_render_pt_device_noalloc call is in flight on this GPU: they share one scratch array");
  DeviceGuard guard(s->device);
  hipStream_t st = s->stream;
  // Every call on this GPU shares the one array: the counters must start at zero (the front
  // end clears them); every partial sum and per-pixel record this render reads is written
  // by it first, so what an earlier, larger render left behind is never looked at.
  const RenderBuffers rb = scratch_buffers(s->noalloc_scratch);
  EventSpans ev;  // off: records nothing, creates nothing
  if (int rc = enqueue_front(s, pl, rb, st, ev)) return rc;
  if (int rc = enqueue_fused(s, pl, rb, st, ev)) return rc;
  if (int rc = enqueue_resolve(pl, rb, d_rgb, st, ev)) return rc;
  HIP_TRY(hipStreamSynchronize(st));
  if (stats) {
    // what the host knows; ray and vertex counts stay on the device (reading them back
    // would need a staging buffer)
    stats->samples = (uint64_t)pl.a.n_pix * pl.a.spp;
    stats->iterations = 1;
    stats->chunk_spp = pl.a.chunk_spp;
    stats->mode = RTR_MODE_FUSED;
    stats->n_columns = pl.a.n_cols;
    stats->fused_launches = 1;
    stats->trace_launches = 1;
    stats->n_devices = 1;
    stats->ms_total =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    stats->ms_render_min = stats->ms_render_max = stats->ms_total;
  }
  return RTR_OK;
}

// One host thread per scene, as the reference's renderer has one per band of rows
// (src/Render.cpp:667-684); the join is followed by the only traffic between devices.
static int render_multi(rtr_scene *const *scenes, int32_t n_scenes, const rtr_camera *cam,
                        const rtr_render_params *prm, float *d_rgb, float *h_rgb, rtr_stats *stats) {
  if (!scenes || !cam || !prm || (!d_rgb && !h_rgb)) return fail(RTR_ERR_INVALID, "null argument");
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_scenes < 1 || n_scenes > kMaxParts) return fail(RTR_ERR_INVALID, "n_scenes must be 1..64");
  for (int r = 0; r < n_scenes; r++) {
    if (!scenes[r]) return fail(RTR_ERR_INVALID, "null scene handle");
    for (int q = 0; q < r; q++)
      if (scenes[q] == scenes[r])
        return fail(RTR_ERR_INVALID, "the same scene handle twice (a handle serves one thread at a time)");
  }
  if (prm->part_count > 1) return fail(RTR_ERR_INVALID, "part_count must be 0 or 1: this call partitions the image itself");
  auto t_start = std::chrono::steady_clock::now();
  const int R = n_scenes;
  rtr_scene *root = scenes[0];
  rtr_render_params base = *prm;
  base.part_count = R;
  base.tile_cols = prm->tile_cols > 0 ? prm->tile_cols : balanced_tile_cols(prm->width, R);
  // validate once, on the calling thread, so that argument errors are plain errors
  {
    RenderPlan pl;
    base.part_index = 0;
    if (int rc = plan_render(root, cam, &base, (size_t)1 << 27, pl)) return rc;
  }
  const int W = prm->width, H = prm->height;
  PartOffsets po{};
  std::vector<int> n_cols(R);
  size_t total = 0;
  for (int r = 0; r < R; r++) {
    n_cols[r] = count_columns(W, r, R, base.tile_cols);
    po.off[r] = total;
    total += (size_t)n_cols[r] * H * 3;
  }
  if (total != (size_t)W * H * 3) return fail(RTR_ERR_DEVICE, "internal: partitions do not cover the image");

  // device 0 of the call holds every partition's framebuffer, then the whole image
  {
    DeviceGuard guard(root->device);
    HIP_TRY(root->ws.gather.reserve(total));
    if (!d_rgb) HIP_TRY(root->ws.rgb.reserve(total));
    for (int r = 1; r < R; r++) enable_peer_access(root->device, scenes[r]->device);
  }
  float *const gathered = root->ws.gather.ptr;
  float *const image = d_rgb ? d_rgb : root->ws.rgb.ptr;

  struct Rank {
    int rc = RTR_OK;
    std::string error;
    rtr_stats st{};
    hipEvent_t arrived = nullptr;
  };
  std::vector<Rank> ranks(R);
  auto work = [&](int r) {
    Rank &me = ranks[r];
    rtr_scene *s = scenes[r];
    auto bail = [&](int rc) {
      me.rc = rc;
      me.error = g_error;
    };
    rtr_render_params p = base;
    p.part_index = r;
    if (n_cols[r] == 0) return;
    const size_t n = (size_t)n_cols[r] * H * 3;
    DeviceGuard guard(s->device);
    if (!guard.ok) return bail(fail(RTR_ERR_DEVICE, "hipSetDevice failed"));
    // rank 0 renders straight into its place; the others into their own framebuffer,
    // which then travels to device 0 in one copy on the rank's own stream
    float *dst = gathered + po.off[r];
    if (r != 0) {
      hipError_t e = s->ws.rgb.reserve(n);
      if (e != hipSuccess) return bail(fail(RTR_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)));
      dst = s->ws.rgb.ptr;
    }
    if (int rc = rtr_render_pt_device(s, cam, &p, dst, &me.st)) return bail(rc);
    if (r != 0) {
      hipError_t e = hipMemcpyPeerAsync(gathered + po.off[r], root->device, dst, s->device,
                                        n * sizeof(float), s->stream);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&me.arrived, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(me.arrived, s->stream);
      if (e != hipSuccess) return bail(fail(RTR_ERR_DEVICE, std::string("gather: ") + hipGetErrorString(e)));
    }
  };
  {
    std::vector<std::thread> threads;
    for (int r = 1; r < R; r++) threads.emplace_back(work, r);
    work(0);
    for (auto &t : threads) t.join();
  }
  auto t_joined = std::chrono::steady_clock::now();
  int rc = RTR_OK;
  {
    DeviceGuard guard(root->device);
    hipError_t e = hipSuccess;
    for (int r = 0; r < R; r++) {
      if (ranks[r].rc != RTR_OK && rc == RTR_OK) rc = fail(ranks[r].rc, "partition " + std::to_string(r) + ": " + ranks[r].error);
      if (ranks[r].arrived && e == hipSuccess) e = hipStreamWaitEvent(root->stream, ranks[r].arrived, 0);
    }
    if (rc == RTR_OK && e == hipSuccess) {
      const size_t n_pix = (size_t)W * H;
      hipLaunchKernelGGL(k_unpartition, dim3(blocks_for(n_pix)), dim3(kBlock), 0, root->stream, W, H, R,
                         base.tile_cols, po, (const float *)gathered, image);
      e = hipGetLastError();
      if (e == hipSuccess && h_rgb)
        e = hipMemcpyAsync(h_rgb, image, total * sizeof(float), hipMemcpyDeviceToHost, root->stream);
    }
    // also on failure: no copy may still be in flight when the events go away
    hipError_t e2 = hipStreamSynchronize(root->stream);
    for (int r = 1; r < R; r++) {
      DeviceGuard g2(scenes[r]->device);
      (void)hipStreamSynchronize(scenes[r]->stream);
    }
    for (Rank &k : ranks)
      if (k.arrived) (void)hipEventDestroy(k.arrived);
    if (rc == RTR_OK && (e != hipSuccess || e2 != hipSuccess))
      rc = fail(RTR_ERR_DEVICE, std::string("gather: ") + hipGetErrorString(e != hipSuccess ? e : e2));
  }
  if (rc) return rc;
  if (stats) {
    auto t_end = std::chrono::steady_clock::now();
    const rtr_stats &s0 = ranks[0].st;
    stats->chunk_spp = s0.chunk_spp;
    stats->pool_size = s0.pool_size;
    stats->mode = s0.mode;
    stats->n_columns = W;
    stats->n_devices = R;
    for (const Rank &k : ranks) {
      stats->samples += k.st.samples;
      stats->rays += k.st.rays;
      stats->vertices += k.st.vertices;
      stats->trace_launches += k.st.trace_launches;
      stats->fused_launches += k.st.fused_launches;
      stats->boxes_tested += k.st.boxes_tested;
      stats->prims_tested += k.st.prims_tested;
      stats->iterations = std::max(stats->iterations, k.st.iterations);
      // kernel times: the slowest partition's, which is what the image waited for
      stats->ms_trace = std::max(stats->ms_trace, k.st.ms_trace);
      stats->ms_logic = std::max(stats->ms_logic, k.st.ms_logic);
      stats->ms_other = std::max(stats->ms_other, k.st.ms_other);
      stats->ms_fused = std::max(stats->ms_fused, k.st.ms_fused);
    }
    stats->ms_render_min = stats->ms_render_max = ranks[0].st.ms_total;
    for (int r = 0; r < R; r++) {
      if (n_cols[r] == 0) continue;
      stats->ms_render_min = std::min(stats->ms_render_min, ranks[r].st.ms_total);
      stats->ms_render_max = std::max(stats->ms_render_max, ranks[r].st.ms_total);
    }
    stats->ms_gather = std::chrono::duration<double, std::milli>(t_end - t_joined).count();
    stats->ms_total = std::chrono::duration<double, std::milli>(t_end - t_start).count();
  }
  return RTR_OK;
}

int rtr_render_pt_multi(rtr_scene *const *scenes, int32_t n_scenes, const rtr_camera *cam,
                        const rtr_render_params *prm, float *rgb_out, rtr_stats *stats) {
  return render_multi(scenes, n_scenes, cam, prm, nullptr, rgb_out, stats);
}

int rtr_render_pt_multi_device(rtr_scene *const *scenes, int32_t n_scenes, const rtr_camera *cam,
                               const rtr_render_params *prm, float *d_rgb_out, rtr_stats *stats) {
  return render_multi(scenes, n_scenes, cam, prm, d_rgb_out, nullptr, stats);
}

} /* extern "C" */

#ifdef RTR_STREAM_STATS
// development only: read and clear the counters of trace_stream
extern "C" int rtr_debug_stream_stats(unsigned long long *out8) {
  unsigned long long zero[8] = {0};
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(rtr::g_stream_stats), sizeof(zero)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(rtr::g_stream_stats), zero, sizeof(zero)) != hipSuccess) return -1;
  return 0;
}
#endif
