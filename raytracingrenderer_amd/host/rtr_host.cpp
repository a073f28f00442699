// Host-side mirror of the reference's scene API: construction, the BVH build,
// flattening into the C ABI's arrays, and Render on top of the device library.
// See rtr_host.hpp.  No ray is intersected and no sample is shaded on the CPU.
#include "rtr_host.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <iostream>
#include <limits>
#include <map>
#include <new>
#include <stdexcept>
#include <thread>

// ---------------- vectors ----------------
float Vector3f::getLen() const { return sqrtf(x * x + y * y + z * z); }
Vector3f Vector3f::normalize() const {
  float len = getLen();
  return Vector3f(x / len, y / len, z / len);
}
float vec::dotProduct(const Vector3f &a, const Vector3f &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Vector3f vec::crossProduct(const Vector3f &a, const Vector3f &b) {
  return Vector3f(a.y * b.z - a.z * b.y, a.z * b.x - b.z * a.x, a.x * b.y - a.y * b.x);
}

// ---------------- Bound3 (include/Bound3.h:10-22) ----------------
Bound3::Bound3()
    : min(std::numeric_limits<float>::max()), max(std::numeric_limits<float>::lowest()) {}

Bound3 &Bound3::merge(const Bound3 &b) {
  Vector3f lo(min.x < b.min.x ? min.x : b.min.x, min.y < b.min.y ? min.y : b.min.y,
              min.z < b.min.z ? min.z : b.min.z);
  Vector3f hi(max.x > b.max.x ? max.x : b.max.x, max.y > b.max.y ? max.y : b.max.y,
              max.z > b.max.z ? max.z : b.max.z);
  min = lo;
  max = hi;
  return *this;
}

Vector3f Bound3::getCenterPos() const { return min * 0.5f + max * 0.5f; }

// ---------------- Material (include/Material.h:53-65) ----------------
Material::Material(MaterialType _type, Vector3f _emission, Vector3f _kd, Vector3f _ks, float _ior,
                   float _roughness, float _specularExponent)
    : type(_type), emission(_emission), kd(_kd), ks(_ks) {
  ior = (float)(_ior > 0 ? (double)_ior : 1e-5);
  roughness = (float)(_roughness > 0 ? (double)(_roughness < 1.0f ? _roughness : 1.0f) : 1e-5);
  specularExponent = _specularExponent > 1 ? _specularExponent : 1.0f;
}

bool Material::hasEmission() const {
  return (double)emission.x > 1e-4 || (double)emission.y > 1e-4 || (double)emission.z > 1e-4;
}

// ---------------- Sphere (include/Sphere.h:11-21) ----------------
Sphere::Sphere(Material *_material, Vector3f _center, float _radius) : Object(_material) {
  center = _center;
  radius = _radius > 0 ? _radius : 0;
  bound = Bound3(center - Vector3f(radius), center + Vector3f(radius));
  area = (float)(4.0 * MY_PI * radius * radius);
}

// ---------------- Trangle (include/Trangle.h:17-38, src/Trangle.cpp:30-40) ----------------
static float min3(float a, float b, float c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
static float max3(float a, float b, float c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

Trangle::Trangle(Material *_material, Vector3f _p0, Vector3f _p1, Vector3f _p2, Vector3f _normal,
                 Vector2f _uv0, Vector2f _uv1, Vector2f _uv2)
    : Object(_material), p0(_p0), p1(_p1), p2(_p2), uv0(_uv0), uv1(_uv1), uv2(_uv2) {
  Vector3f faceCross = vec::crossProduct(p1 - p0, p2 - p0);
  // a supplied normal is used only when it is (close to) unit length; a NaN one
  // fails the comparison and the geometric normal is taken
  if ((double)(_normal.x * _normal.x + _normal.y * _normal.y + _normal.z * _normal.z) >= 1.0 - 1e-4)
    normal = _normal.normalize();
  else
    normal = faceCross.normalize();
  bound = Bound3(Vector3f(min3(p0.x, p1.x, p2.x), min3(p0.y, p1.y, p2.y), min3(p0.z, p1.z, p2.z)),
                 Vector3f(max3(p0.x, p1.x, p2.x), max3(p0.y, p1.y, p2.y), max3(p0.z, p1.z, p2.z)));
  area = faceCross.getLen();  // twice the geometric area, as in the reference
}

// fn(from, to) over [0, count) in contiguous pieces, on several threads when the range is large
template <class F>
static void parallel_ranges(size_t count, F fn) {
  unsigned workers = std::thread::hardware_concurrency();
  if (workers > 16) workers = 16;
  if (count < (1u << 15) || workers < 2) return fn((size_t)0, count);
  std::vector<std::future<void>> parts;
  for (unsigned w = 0; w < workers; w++)
    parts.push_back(std::async(std::launch::async, fn, count * w / workers, count * (w + 1) / workers));
  for (auto &part : parts) part.wait();
  for (auto &part : parts) part.get();
}

// ---------------- BVH (src/BVH.cpp:19-64) ----------------
void BVH::destroy(BVHNode *n) {
  if (!n) return;
  destroy(n->left);
  destroy(n->right);
  delete n;
}

namespace {

// One record of the array the build sorts: the object and its sort key at the current
// node.  std::sort decides by comparison outcomes and positions only, so sorting these
// records by `key` permutes them exactly as the reference's sort of the object pointers
// with the comparator "centre(a).axis < centre(b).axis" does (src/BVH.cpp:42-55), ties
// included, without recomputing two box centres per comparison.
struct SortRecord {
  float key;
  Object *obj;
};

// Subtrees at least this large are built by a task of their own while tasks remain.
const size_t kParallelBuildMin = 1 << 14;

struct TreeBuilder {
  std::vector<SortRecord> rec;

  // Builds the subtree over rec[lo, hi).  Sorting a sub-range in place gives what the
  // reference's copy of that half into a fresh vector gives (src/BVH.cpp:58-61): the
  // sort sees the same sequence either way.  `top` receives the order after this node's
  // own sort, the only one the reference applies to the caller's vector.
  BVHNode *build(size_t lo, size_t hi, int spare_tasks, std::vector<Object *> *top) {
    BVHNode *node = new BVHNode();
    Bound3 all;
    for (size_t k = lo; k < hi; k++) all.merge(rec[k].obj->getBound());
    node->bound = all;
    if (hi - lo == 1) {
      node->object = rec[lo].obj;
      return node;
    }
    const float ex = all.max.x - all.min.x, ey = all.max.y - all.min.y, ez = all.max.z - all.min.z;
    const int axis = ex > ey ? (ex > ez ? 0 : 2) : (ey > ez ? 1 : 2);
    for (size_t k = lo; k < hi; k++) {
      Vector3f c = rec[k].obj->getBound().getCenterPos();
      rec[k].key = axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
    }
    std::sort(rec.begin() + lo, rec.begin() + hi,
              [](const SortRecord &a, const SortRecord &b) { return a.key < b.key; });
    if (top)
      for (size_t k = lo; k < hi; k++) (*top)[k - lo] = rec[k].obj;
    const size_t mid = lo + (hi - lo) / 2;
    if (spare_tasks > 0 && hi - lo >= kParallelBuildMin) {
      // the two halves touch disjoint ranges of rec
      const int give = (spare_tasks - 1) / 2;
      auto lower = std::async(std::launch::async,
                              [this, lo, mid, give] { return build(lo, mid, give, nullptr); });
      try {
        node->right = build(mid, hi, spare_tasks - 1 - give, nullptr);
      } catch (...) {
        lower.wait();
        throw;
      }
      node->left = lower.get();
    } else {
      node->left = build(lo, mid, 0, nullptr);
      node->right = build(mid, hi, 0, nullptr);
    }
    return node;
  }
};

}  // namespace

BVHNode *BVH::build(std::vector<Object *> &objs) {
  if (objs.empty()) return nullptr;
  TreeBuilder tb;
  tb.rec.resize(objs.size());
  for (size_t k = 0; k < objs.size(); k++) tb.rec[k] = SortRecord{0.f, objs[k]};
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (hw > 32) hw = 32;
  return tb.build(0, objs.size(), (int)hw - 1, &objs);
}

BVH::Builder BVH::builder = BVH::Auto;
size_t BVH::deviceMin = 1 << 13;
int BVH::buildDevice = -1;  // the calling thread's current device

void BVH::clear() {
  if (arena) free(arena);  // BVHNode has no destructor to run
  else destroy(root);
  arena = nullptr;
  root = nullptr;
}

// BVH::selfBuildBVH by the kernels behind rtr_build_bvh: the objects' boxes go over, node arrays
// come back (numbered level by level, node 0 the root) and become BVHNodes in one block.
// Returns null, having changed nothing, when `finite_only` is set and the kernels decline the job:
// some box plane is infinite or not a number, or the sort needs a heap sort too long for one lane.
BVHNode *BVH::buildOnDevice(std::vector<Object *> &objs, bool finite_only) {
  const size_t n = objs.size(), nn = 2 * n - 1;
  const bool times = getenv("RTR_TREE_TIMES") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!times) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "BVH::buildOnDevice %-12s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  BulkVector<float> ob[6], nb[6];
  for (int a = 0; a < 6; a++) ob[a].resize(n), nb[a].resize(nn);
  std::atomic<bool> finite{true};
  parallel_ranges(n, [&](size_t from, size_t to) {
    bool ok = true;
    for (size_t k = from; k < to; k++) {
      const Bound3 b = objs[k]->getBound();
      ob[0][k] = b.min.x, ob[1][k] = b.min.y, ob[2][k] = b.min.z;
      ob[3][k] = b.max.x, ob[4][k] = b.max.y, ob[5][k] = b.max.z;
      for (int a = 0; a < 6; a++) ok = ok && std::isfinite(ob[a][k]);
    }
    if (!ok) finite = false;
  });
  if (finite_only && !finite) return nullptr;
  BulkVector<int32_t> left(nn), right(nn), object(nn), order(n);
  const float *omin[3] = {ob[0].data(), ob[1].data(), ob[2].data()};
  const float *omax[3] = {ob[3].data(), ob[4].data(), ob[5].data()};
  float *nmin[3] = {nb[0].data(), nb[1].data(), nb[2].data()};
  float *nmax[3] = {nb[3].data(), nb[4].data(), nb[5].data()};
  // first touch of the result arrays by several threads (the copy from the device would otherwise
  // fault every page in on one)
  parallel_ranges(nn, [&](size_t from, size_t to) {
    for (size_t i = from; i < to; i += 1024) {
      for (int a = 0; a < 6; a++) nb[a][i] = 0.f;
      left[i] = right[i] = object[i] = 0;
    }
  });
  mark("boxes");
  const int rc = rtr_build_bvh(buildDevice, (int32_t)n, omin, omax, nmin, nmax, left.data(), right.data(),
                               object.data(), order.data());
  if (rc == RTR_ERR_UNSUPPORTED && finite_only) return nullptr;  // a sort the kernels leave to the host
  if (rc != RTR_OK) throw std::runtime_error(std::string("rtr_build_bvh: ") + rtr_last_error());
  mark("rtr_build_bvh");
  // raw storage, first touched by the threads that fill it
  arena = static_cast<BVHNode *>(malloc(nn * sizeof(BVHNode)));
  if (!arena) throw std::bad_alloc();
  parallel_ranges(nn, [&](size_t from, size_t to) {
    for (size_t i = from; i < to; i++) {
      BVHNode &node = *new (&arena[i]) BVHNode();
      node.bound = Bound3(Vector3f(nb[0][i], nb[1][i], nb[2][i]), Vector3f(nb[3][i], nb[4][i], nb[5][i]));
      node.left = left[i] >= 0 ? &arena[left[i]] : nullptr;
      node.right = right[i] >= 0 ? &arena[right[i]] : nullptr;
      node.object = object[i] >= 0 ? objs[object[i]] : nullptr;
    }
  });
  std::vector<Object *> sorted(n);
  parallel_ranges(n, [&](size_t from, size_t to) {
    for (size_t k = from; k < to; k++) sorted[k] = objs[order[k]];
  });
  objs.swap(sorted);
  mark("nodes");
  return &arena[0];
}

void BVH::buildBVH(std::vector<Object *> &objs) {
  clear();
  Builder how = builder;
  if (const char *env = getenv("RTR_TREE_BUILDER")) {
    if (!strcmp(env, "host")) how = Host;
    else if (!strcmp(env, "device")) how = Device;
  }
  bool on_device = how == Device && !objs.empty();
  if (how == Auto && objs.size() >= deviceMin) {
    int devices = 0;
    on_device = rtr_device_count(&devices) == RTR_OK && devices > (buildDevice < 0 ? 0 : buildDevice);
  }
  if (on_device) root = buildOnDevice(objs, how == Auto);  // forced: a bad box is the call's error
  if (!root) root = build(objs);
}

// ---------------- TrangleMesh (include/Trangle.h:60-95) ----------------
TrangleMesh::TrangleMesh(Material *_material, std::vector<Vector3f> *_positions,
                         std::vector<unsigned int> *_posIndices, std::vector<Vector3f> *_normals,
                         std::vector<Vector2f> *_textureCoords)
    : Object(_material), bvh(nullptr) {
  if (!_positions || !_posIndices) throw std::invalid_argument("TrangleMesh: null positions/indices");
  if (_posIndices->size() % 3 != 0) throw std::invalid_argument("TrangleMesh: indices are not triples");
  const std::vector<Vector3f> &P = *_positions;
  const std::vector<unsigned int> &I = *_posIndices;
  trangles.resize(I.size() / 3);
  area = 0;
  for (size_t k = 0; k < I.size(); k++)
    if (I[k] >= P.size()) throw std::out_of_range("TrangleMesh: vertex index out of range");
  // The triangles do not depend on one another: large meshes construct them on several threads.
  // The mesh's area is summed afterwards, in index order, as the reference's loop does
  // (include/Trangle.h:79-91).
  const size_t count = I.size() / 3;
  // one block for all triangles instead of one allocation each: a million small allocations from
  // sixteen threads spend their time in the allocator and in page faults
  block = static_cast<Trangle *>(malloc((count ? count : 1) * sizeof(Trangle)));
  if (!block) throw std::bad_alloc();
  auto construct = [&](size_t from, size_t to) {
    for (size_t f = from; f < to; f++) {
      const unsigned a = I[3 * f], b = I[3 * f + 1], c = I[3 * f + 2];
      Vector3f n(0);
      Vector2f uv[3];
      if (_normals) n = (*_normals)[a] + (*_normals)[b] + (*_normals)[c];
      if (_textureCoords) {
        uv[0] = (*_textureCoords)[a];
        uv[1] = (*_textureCoords)[b];
        uv[2] = (*_textureCoords)[c];
      }
      // without normals this passes normalize(0) = NaN, which selects the face normal
      Trangle *t = new (block + f) Trangle(_material, P[a], P[b], P[c], n.normalize(), uv[0], uv[1], uv[2]);
      t->indexInMesh = (int)f;
      trangles[f] = t;
    }
  };
  parallel_ranges(count, construct);
  for (size_t f = 0; f < count; f++) area += trangles[f]->getArea();
  bvh = new BVH();
  bvh->buildBVH(trangles);
  if (!trangles.empty()) bound = bvh->getBound();
}

TrangleMesh::~TrangleMesh() {
  delete bvh;
  for (Object *t : trangles) static_cast<Trangle *>(t)->~Trangle();
  free(block);
}

// ---------------- Scene (include/Scene.h:33-46) ----------------
static void rtr_check(int rc, const char *what) {
  if (rc != RTR_OK) throw std::runtime_error(std::string(what) + ": " + rtr_last_error());
}

Scene::Scene(Camera *_camera) : camera(_camera), bvh(new BVH()) {}

Scene::~Scene() {
  if (device) rtr_scene_destroy(device);
  dropReplicas();
  delete bvh;
}

void Scene::addObject(Object *obj) {
  objects.push_back(obj);
  addOrder.push_back(obj);
  if (obj->getMaterial()->hasEmission()) lights.push_back(obj);
  flatValid = false;
}

void Scene::rebuildBVH() {
  bvh->buildBVH(objects);
  flatValid = false;
}

void Scene::setDevice(int index) {
  if (index != deviceIndex && device) {
    rtr_scene_destroy(device);
    device = nullptr;
  }
  deviceIndex = index;
}

namespace {

struct Flattener {
  FlatScene &f;
  std::map<const Object *, int> prim_of;    // spheres: their primitive id
  std::map<const Object *, int> mesh_base;  // meshes: primitive id of their first triangle
  std::map<const Material *, int> mat_of;

  // a mesh's triangles take consecutive ids in the order of its index triples
  int triangle_prim(const Object *mesh, const Trangle *t) const {
    return mesh_base.at(mesh) + t->indexInMesh;
  }

  int material_id(const Material *m) {
    auto it = mat_of.find(m);
    if (it != mat_of.end()) return it->second;
    int id = (int)f.mat_kind.size();
    mat_of[m] = id;
    f.mat_kind.push_back((int)m->getMaterialType());
    Vector3f e = m->getEmission(), kd = m->getKd(), ks = m->getKs();
    const float ev[3] = {e.x, e.y, e.z}, kdv[3] = {kd.x, kd.y, kd.z}, ksv[3] = {ks.x, ks.y, ks.z};
    for (int a = 0; a < 3; a++) {
      f.mat_emission[a].push_back(ev[a]);
      f.mat_kd[a].push_back(kdv[a]);
      f.mat_ks[a].push_back(ksv[a]);
    }
    f.mat_ior.push_back(m->getIor());
    f.mat_roughness.push_back(m->getRoughness());
    return id;
  }

  void add_sphere(const Sphere *s) {
    int prim = (int)f.prim_kind.size();
    prim_of[s] = prim;
    f.prim_kind.push_back(RTR_PRIM_SPHERE);
    f.prim_index.push_back((int)f.sph_r.size());
    f.prim_material.push_back(material_id(s->getMaterial()));
    f.prim_object.push_back(const_cast<Sphere *>(s));
    Vector3f c = s->getCenter();
    f.sph_c[0].push_back(c.x);
    f.sph_c[1].push_back(c.y);
    f.sph_c[2].push_back(c.z);
    f.sph_r.push_back(s->getRadius());
    f.sph_area.push_back(s->getArea());
  }

  void collect_triangles(const BVHNode *n, std::vector<const Trangle *> &out) {
    if (!n) return;
    if (n->object) {
      out.push_back(static_cast<const Trangle *>(n->object));
    } else {
      collect_triangles(n->left, out);
      collect_triangles(n->right, out);
    }
  }

  void add_mesh(const TrangleMesh *m) {
    // prim ids follow the index-triple order; every triangle knows its place in it
    const std::vector<Object *> &tris = m->getTrangles();
    const size_t count = tris.size(), prim0 = f.prim_kind.size(), row0 = f.tri_area.size();
    const int mat = material_id(m->getMaterial());
    mesh_base[m] = (int)prim0;
    f.prim_kind.resize(prim0 + count, RTR_PRIM_TRIANGLE);
    f.prim_index.resize(prim0 + count);
    f.prim_material.resize(prim0 + count, mat);
    f.prim_object.resize(prim0 + count);
    f.tri_area.resize(row0 + count);
    for (int a = 0; a < 3; a++) {
      f.tri_p0[a].resize(row0 + count), f.tri_p1[a].resize(row0 + count);
      f.tri_p2[a].resize(row0 + count), f.tri_n[a].resize(row0 + count);
    }
    parallel_ranges(count, [&](size_t from, size_t to) {
      for (size_t k = from; k < to; k++) {
        const Trangle *t = static_cast<const Trangle *>(tris[k]);
        const size_t i = (size_t)t->indexInMesh, row = row0 + i;
        f.prim_index[prim0 + i] = (int)row;
        f.prim_object[prim0 + i] = const_cast<Trangle *>(t);
        const Vector3f *p[3] = {&t->P0(), &t->P1(), &t->P2()};
        BulkVector<float> *dst[3] = {f.tri_p0, f.tri_p1, f.tri_p2};
        for (int v = 0; v < 3; v++) dst[v][0][row] = p[v]->x, dst[v][1][row] = p[v]->y, dst[v][2][row] = p[v]->z;
        f.tri_n[0][row] = t->getNormal().x, f.tri_n[1][row] = t->getNormal().y, f.tri_n[2][row] = t->getNormal().z;
        f.tri_area[row] = t->getArea();
      }
    });
  }

  // One entry per node that the walk below visits.
  struct Visit {
    const BVHNode *node;
    const Object *mesh;  // the mesh whose nested tree the node belongs to, null in the top-level tree
    int id, right;       // its number; number of the right child, -1 for a leaf
  };
  // A mesh whose tree is one block of nodes in depth-first order already (built on the device):
  // its nodes take consecutive numbers and are copied without a walk.
  struct Block {
    const BVHNode *nodes;
    const Object *mesh;
    int first, count;
  };

  // Numbers the nodes depth first, left child first, with an explicit stack, then fills the arrays
  // on several threads.  A mesh leaf of the top-level tree is replaced by the mesh's own tree: its
  // root carries the same bound (include/Trangle.h:56), so the second box test the reference
  // performs there cannot change the outcome.
  void add_tree(const BVHNode *root) {
    std::vector<Visit> order;
    std::vector<Block> blocks;
    struct Pending {
      const BVHNode *node;
      const Object *mesh;
      int parent;  // entry of `order` whose right child this is, -1: none
    };
    std::vector<Pending> stack{{root, nullptr, -1}};
    int next_id = 0;
    while (!stack.empty()) {
      Pending at = stack.back();
      stack.pop_back();
      const int id = next_id;
      if (at.parent >= 0) order[at.parent].right = id;
      if (!at.mesh && at.node->object && at.node->object->kind() == Object::MESH) {
        const TrangleMesh *m = static_cast<const TrangleMesh *>(at.node->object);
        const BVHNode *inner = m->getBVH()->getRoot();
        if (inner && m->getBVH()->preorderBlock()) {
          const int count = 2 * (int)m->numTrangles() - 1;
          blocks.push_back(Block{inner, m, id, count});
          next_id += count;
          continue;
        }
        if (inner) at.mesh = m, at.node = inner;
      }
      next_id++;
      order.push_back(Visit{at.node, at.mesh, id, -1});
      if (!at.node->object) {
        stack.push_back(Pending{at.node->right, at.mesh, (int)order.size() - 1});
        stack.push_back(Pending{at.node->left, at.mesh, -1});
      }
    }
    const size_t count = (size_t)next_id;
    for (int a = 0; a < 3; a++) f.node_bmin[a].resize(count), f.node_bmax[a].resize(count);
    f.node_left.resize(count), f.node_right.resize(count), f.node_prim.resize(count);
    auto put = [&](size_t id, const BVHNode *n, const Object *mesh, int left, int right) {
      f.node_bmin[0][id] = n->bound.min.x, f.node_bmin[1][id] = n->bound.min.y, f.node_bmin[2][id] = n->bound.min.z;
      f.node_bmax[0][id] = n->bound.max.x, f.node_bmax[1][id] = n->bound.max.y, f.node_bmax[2][id] = n->bound.max.z;
      f.node_left[id] = left, f.node_right[id] = right;
      f.node_prim[id] = !n->object ? -1
                        : mesh     ? triangle_prim(mesh, static_cast<const Trangle *>(n->object))
                                   : prim_of.at(n->object);
    };
    parallel_ranges(order.size(), [&](size_t from, size_t to) {
      for (size_t k = from; k < to; k++) {
        const Visit &v = order[k];
        put((size_t)v.id, v.node, v.mesh, v.node->object ? -1 : v.id + 1, v.right);
      }
    });
    for (const Block &b : blocks) {
      const int base = mesh_base.at(b.mesh);
      parallel_ranges((size_t)b.count, [&](size_t from, size_t to) {
        for (size_t i = from; i < to; i++) {
          const BVHNode *n = b.nodes + i;
          const size_t id = (size_t)b.first + i;
          f.node_bmin[0][id] = n->bound.min.x, f.node_bmin[1][id] = n->bound.min.y, f.node_bmin[2][id] = n->bound.min.z;
          f.node_bmax[0][id] = n->bound.max.x, f.node_bmax[1][id] = n->bound.max.y, f.node_bmax[2][id] = n->bound.max.z;
          f.node_left[id] = n->left ? b.first + (int)(n->left - b.nodes) : -1;
          f.node_right[id] = n->right ? b.first + (int)(n->right - b.nodes) : -1;
          f.node_prim[id] = n->object ? base + static_cast<const Trangle *>(n->object)->indexInMesh : -1;
        }
      });
    }
  }
};

}  // namespace

namespace {
// the depth-first numbering of Flattener::add_tree, looking for flattened node `want`
bool set_bound(BVHNode *n, int &next, int want, const Bound3 &b) {
  if (n->object && n->object->kind() == Object::MESH) {
    BVHNode *inner = const_cast<BVHNode *>(static_cast<const TrangleMesh *>(n->object)->getBVH()->getRoot());
    if (inner) {
      if (next == want) n->bound = b;  // the top-level leaf and the mesh's root are one flattened node
      return set_bound(inner, next, want, b);
    }
  }
  if (next++ == want) {
    n->bound = b;
    return true;
  }
  if (n->object) return false;
  return set_bound(n->left, next, want, b) || set_bound(n->right, next, want, b);
}
}  // namespace

bool Scene::setNodeBound(int flatIndex, const Bound3 &b) {
  BVHNode *root = const_cast<BVHNode *>(bvh->getRoot());
  if (!root || flatIndex < 0) return false;
  int next = 0;
  if (!set_bound(root, next, flatIndex, b)) return false;
  flatValid = false;
  return true;
}

rtr_scene_desc FlatScene::desc() const {
  rtr_scene_desc d;
  memset(&d, 0, sizeof(d));
  d.n_nodes = (int32_t)node_prim.size();
  d.root = 0;
  for (int a = 0; a < 3; a++) {
    d.node_bmin[a] = node_bmin[a].data();
    d.node_bmax[a] = node_bmax[a].data();
    d.tri_p0[a] = tri_p0[a].data();
    d.tri_p1[a] = tri_p1[a].data();
    d.tri_p2[a] = tri_p2[a].data();
    d.tri_n[a] = tri_n[a].data();
    d.sph_c[a] = sph_c[a].data();
    d.mat_emission[a] = mat_emission[a].data();
    d.mat_kd[a] = mat_kd[a].data();
    d.mat_ks[a] = mat_ks[a].data();
  }
  d.node_left = node_left.data();
  d.node_right = node_right.data();
  d.node_prim = node_prim.data();
  d.n_prims = (int32_t)prim_kind.size();
  d.prim_kind = prim_kind.data();
  d.prim_index = prim_index.data();
  d.prim_material = prim_material.data();
  d.n_tris = (int32_t)tri_area.size();
  d.tri_area = tri_area.data();
  d.n_spheres = (int32_t)sph_r.size();
  d.sph_r = sph_r.data();
  d.sph_area = sph_area.data();
  d.n_materials = (int32_t)mat_kind.size();
  d.mat_kind = mat_kind.data();
  d.mat_ior = mat_ior.data();
  d.mat_roughness = mat_roughness.data();
  d.n_lights = (int32_t)light_kind.size();
  d.light_kind = light_kind.data();
  d.light_prim = light_prim.data();
  d.light_area = light_area.data();
  d.light_tri_begin = light_tri_begin.data();
  d.light_tris = light_tris.data();
  return d;
}

const FlatScene &Scene::flatten() {
  if (flatValid) return flat;
  flat = FlatScene();
  Flattener fl{flat, {}, {}};
  for (Object *o : addOrder) {
    if (o->kind() == Object::SPHERE) fl.add_sphere(static_cast<Sphere *>(o));
    else if (o->kind() == Object::MESH) fl.add_mesh(static_cast<TrangleMesh *>(o));
    else throw std::invalid_argument("Scene: only Sphere and TrangleMesh can be added");
  }
  if (bvh->getRoot()) fl.add_tree(bvh->getRoot());
  flat.light_tri_begin.push_back(0);
  for (Object *l : lights) {
    flat.light_area.push_back(l->getArea());
    if (l->kind() == Object::SPHERE) {
      flat.light_kind.push_back(RTR_PRIM_SPHERE);
      flat.light_prim.push_back(fl.prim_of.at(l));
    } else {
      flat.light_kind.push_back(RTR_PRIM_TRIANGLE);
      flat.light_prim.push_back(-1);
      std::vector<const Trangle *> leaf_order;
      fl.collect_triangles(static_cast<TrangleMesh *>(l)->getBVH()->getRoot(), leaf_order);
      for (const Trangle *t : leaf_order) flat.light_tris.push_back(fl.triangle_prim(l, t));
    }
    flat.light_tri_begin.push_back((int32_t)flat.light_tris.size());
  }
  flatValid = true;
  if (device) {
    rtr_scene_destroy(device);
    device = nullptr;
  }
  dropReplicas();
  return flat;
}

void Scene::dropReplicas() {
  for (rtr_scene *r : replicas) rtr_scene_destroy(r);
  replicas.clear();
  replicaDevices.clear();
}

const std::vector<rtr_scene *> &Scene::deviceScenes(const std::vector<int> &devices) {
  flatten();
  if (devices != replicaDevices) {
    dropReplicas();
    rtr_scene_desc d = flat.desc();
    for (int dev : devices) {
      rtr_scene *r = nullptr;
      rtr_check(rtr_scene_create(&d, dev, &r), "rtr_scene_create");
      replicas.push_back(r);
      replicaDevices.push_back(dev);
    }
  }
  return replicas;
}

rtr_scene *Scene::deviceScene() {
  flatten();
  if (!device) {
    rtr_scene_desc d = flat.desc();
    rtr_check(rtr_scene_create(&d, deviceIndex, &device), "rtr_scene_create");
  }
  return device;
}

void Scene::intersectBatch(const float *org_xyz, const float *dir_xyz, size_t n, int32_t *prim_id,
                           float *t, float *pos_xyz, float *nrm_xyz) {
  rtr_check(rtr_trace_closest(deviceScene(), org_xyz, dir_xyz, n, prim_id, t, pos_xyz, nrm_xyz),
            "rtr_trace_closest");
}

void Scene::checkConnectableBatch(const float *a_xyz, const float *b_xyz, size_t n, uint8_t *connected) {
  rtr_check(rtr_check_connectable(deviceScene(), a_xyz, b_xyz, n, connected), "rtr_check_connectable");
}

bool Scene::checkConnectable(Vector3f a, Vector3f b, float) {
  const float pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z};
  uint8_t out = 0;
  checkConnectableBatch(pa, pb, 1, &out);
  return out != 0;
}

Intersection Scene::intersect(const Ray &ray) {
  const float o[3] = {ray.startPos.x, ray.startPos.y, ray.startPos.z};
  const float d[3] = {ray.dir.x, ray.dir.y, ray.dir.z};
  int32_t prim = -1;
  float t = 0, pos[3], nrm[3];
  intersectBatch(o, d, 1, &prim, &t, pos, nrm);
  Intersection it;
  if (prim >= 0) {
    it.hit = true;
    it.distance = t;
    it.postion = Vector3f(pos[0], pos[1], pos[2]);
    it.normal = Vector3f(nrm[0], nrm[1], nrm[2]);
    it.object = flat.prim_object[prim];
  }
  return it;
}

// ---------------- Render (src/Render.cpp:3-21, 632-694) ----------------
static float clamp01(float v) { return std::max(0.0f, std::min(1.0f, v)); }

void Render::saveFile(const std::vector<Vector3f> &pixels, const std::string &fileName,
                      unsigned int height, unsigned int width) {
  FILE *fp = fopen((fileName + ".ppm").c_str(), "wb");
  if (!fp) throw std::runtime_error("Render::saveFile: cannot open " + fileName + ".ppm");
  // the reference prints height before width, so viewers show the image transposed
  fprintf(fp, "P6\n%d %d\n255\n", height, width);
  std::vector<unsigned char> row(3 * (size_t)height * width);
  for (size_t i = 0; i < (size_t)height * width; i++) {
    row[3 * i] = (unsigned char)(255 * powf(clamp01(pixels[i].x), 0.6f));
    row[3 * i + 1] = (unsigned char)(255 * powf(clamp01(pixels[i].y), 0.6f));
    row[3 * i + 2] = (unsigned char)(255 * powf(clamp01(pixels[i].z), 0.6f));
  }
  fwrite(row.data(), 1, row.size(), fp);
  fclose(fp);
}

void Render::renderPathTracing(Scene &scene, const unsigned int &imageWidth,
                               const unsigned int &imageHeight, const unsigned int &spp,
                               std::string fileName) {
  std::cout << "render start, spp = " << spp << "\n";
  auto start = std::chrono::steady_clock::now();
  const Camera *c = scene.getCamera();
  rtr_camera cam;
  const Vector3f *src[3] = {&c->position, &c->frontDir, &c->upDir};
  float *dst[3] = {cam.position, cam.front, cam.up};
  for (int k = 0; k < 3; k++) {
    dst[k][0] = src[k]->x;
    dst[k][1] = src[k]->y;
    dst[k][2] = src[k]->z;
  }
  cam.fov = c->fov;
  rtr_render_params p;
  memset(&p, 0, sizeof(p));
  p.width = (int32_t)imageWidth;
  p.height = (int32_t)imageHeight;
  p.spp = (int32_t)spp;
  p.seed = options.seed;
  p.max_depth = options.max_depth;
  p.rr_prob = options.rr_prob;
  p.chunk_spp = options.chunk_spp;
  p.pool_size = options.pool_size;
  p.part_index = options.part_index;
  p.part_count = options.part_count;
  p.tile_cols = options.tile_cols;
  p.profile = options.profile ? 1 : 0;
  p.mode = options.mode;
  p.streams = options.streams;
  static_assert(sizeof(Vector3f) == 3 * sizeof(float), "Vector3f must be three packed floats");
  int32_t n_cols = 0;
  if (!options.devices.empty()) {
    // several GPUs: the library partitions, renders concurrently and gathers
    if (options.part_count > 1)
      throw std::invalid_argument("RenderOptions: devices and part_count are exclusive");
    n_cols = p.width;
    p.part_index = 0;
    p.part_count = 0;
    const std::vector<rtr_scene *> &scenes = scene.deviceScenes(options.devices);
    delete buffer;
    buffer = new std::vector<Vector3f>((size_t)n_cols * imageHeight);
    rtr_check(rtr_render_pt_multi(scenes.data(), (int32_t)scenes.size(), &cam, &p,
                                  reinterpret_cast<float *>(buffer->data()), &lastStats),
              "rtr_render_pt_multi");
  } else {
    rtr_check(rtr_partition_columns(p.width, p.part_index, p.part_count, p.tile_cols, &n_cols, nullptr),
              "rtr_partition_columns");
    delete buffer;
    buffer = new std::vector<Vector3f>((size_t)n_cols * imageHeight);
    rtr_check(rtr_render_pt(scene.deviceScene(), &cam, &p, reinterpret_cast<float *>(buffer->data()),
                            &lastStats),
              "rtr_render_pt");
  }
  // src/Render.cpp:669-684: with fewer than 16 rows the reference's row bands are empty
  // (step = height / 16 = 0) and its one thread renders row 0 only; the other rows keep the
  // zeros the buffer was created with.  The device renders every row; the rule is applied here.
  if (imageHeight < 16)
    for (size_t i = 0; i < (size_t)n_cols; i++)
      for (size_t j = 1; j < imageHeight; j++) (*buffer)[i * imageHeight + j] = Vector3f();
  if (!fileName.empty()) saveFile(*buffer, fileName, imageHeight, (unsigned)n_cols);
  std::cout << "render end, spp = " << spp << "\n";
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
  std::cout << "Render complete: \nTime taken: " << ms << " ms\n";
}

void Render::render(Scene &scene, const unsigned int &imageWidth, const unsigned int &imageHeight,
                    const unsigned int &spp, std::string fileName, RenderType type) {
  if (type != RenderType::PathTracing) throw std::invalid_argument("Render::render: unknown RenderType");
  renderPathTracing(scene, imageWidth, imageHeight, spp, fileName);
}

void Render::createPhotonMap(Scene &scene, std::vector<Photon *> *photonMap, unsigned int photonNum) {
  if (!photonMap) throw std::invalid_argument("Render::createPhotonMap: null photon map");
  photonMap->assign(photonNum, nullptr);
  if (photonNum == 0) return;
  std::vector<float> pos(3 * (size_t)photonNum), wi(pos.size()), power(pos.size());
  uint64_t stored = 0;
  rtr_check(rtr_create_photon_map(scene.deviceScene(), photonNum, options.seed, pos.data(), wi.data(),
                                  power.data(), &stored, nullptr),
            "rtr_create_photon_map");
  for (uint64_t k = 0; k < stored; k++) {
    Photon *ph = new Photon();
    ph->pos = Vector3f(pos[3 * k], pos[3 * k + 1], pos[3 * k + 2]);
    ph->wi = Vector3f(wi[3 * k], wi[3 * k + 1], wi[3 * k + 2]);
    ph->power = Vector3f(power[3 * k], power[3 * k + 1], power[3 * k + 2]);
    (*photonMap)[k] = ph;
  }
}

// ---------------- C facade for language bindings (tests drive it with ctypes) ----------------
namespace {
struct HostScene {
  Camera *camera = nullptr;
  Scene *scene = nullptr;
  std::vector<Material *> mats;
  std::vector<Object *> objs;
  Render render;
  rtr_scene_desc desc{};
  std::string error;
  ~HostScene() {
    delete scene;
    for (Object *o : objs) delete o;
    for (Material *m : mats) delete m;
    delete camera;
  }
};
thread_local std::string g_host_error;
template <class F>
int guarded(F &&fn) {
  try {
    fn();
    return 0;
  } catch (const std::exception &e) {
    g_host_error = e.what();
    return -1;
  }
}
Vector3f v3(const float *p) { return Vector3f(p[0], p[1], p[2]); }
}  // namespace

extern "C" {

const char *rtrh_last_error(void) { return g_host_error.c_str(); }

void *rtrh_scene_new(const float pos[3], const float front[3], const float up[3], float fov) {
  HostScene *h = new HostScene();
  h->camera = new Camera(v3(pos), v3(front), v3(up), fov);
  h->scene = new Scene(h->camera);
  return h;
}

void rtrh_scene_free(void *hs) { delete (HostScene *)hs; }

int rtrh_add_material(void *hs, int kind, const float e[3], const float kd[3], const float ks[3],
                      float ior, float roughness) {
  HostScene *h = (HostScene *)hs;
  h->mats.push_back(new Material((MaterialType)kind, v3(e), v3(kd), v3(ks), ior, roughness));
  return (int)h->mats.size() - 1;
}

int rtrh_add_sphere(void *hs, int material, const float c[3], float radius) {
  HostScene *h = (HostScene *)hs;
  int id = -1;
  guarded([&] {
    Sphere *s = new Sphere(h->mats.at(material), v3(c), radius);
    h->objs.push_back(s);
    h->scene->addObject(s);
    id = (int)h->objs.size() - 1;
  });
  return id;
}

int rtrh_add_mesh(void *hs, int material, const float *positions, int n_pos, const uint32_t *idx,
                  int n_idx) {
  HostScene *h = (HostScene *)hs;
  int id = -1;
  guarded([&] {
    std::vector<Vector3f> pos(n_pos);
    for (int i = 0; i < n_pos; i++) pos[i] = v3(positions + 3 * i);
    std::vector<unsigned int> ind(idx, idx + n_idx);
    TrangleMesh *m = new TrangleMesh(h->mats.at(material), &pos, &ind);
    h->objs.push_back(m);
    h->scene->addObject(m);
    id = (int)h->objs.size() - 1;
  });
  return id;
}

int rtrh_build(void *hs) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] { h->scene->rebuildBVH(); });
}

// tree in the flattener's numbering (depth-first, left first); NULL arrays -> count only
int rtrh_flatten(void *hs, float *bmin3, float *bmax3, int32_t *child2, int32_t *prim) {
  HostScene *h = (HostScene *)hs;
  int n = -1;
  guarded([&] {
    const FlatScene &f = h->scene->flatten();
    n = (int)f.node_prim.size();
    for (int i = 0; i < n; i++) {
      for (int a = 0; a < 3; a++) {
        if (bmin3) bmin3[3 * i + a] = f.node_bmin[a][i];
        if (bmax3) bmax3[3 * i + a] = f.node_bmax[a][i];
      }
      if (child2) {
        child2[2 * i] = f.node_left[i];
        child2[2 * i + 1] = f.node_right[i];
      }
      if (prim) prim[i] = f.node_prim[i];
    }
  });
  return n;
}

int rtrh_num_prims(void *hs) {
  HostScene *h = (HostScene *)hs;
  int n = -1;
  guarded([&] { n = (int)h->scene->flatten().prim_kind.size(); });
  return n;
}

float rtrh_object_area(void *hs, int object) {
  HostScene *h = (HostScene *)hs;
  return h->objs.at(object)->getArea();
}

// light table as handed to the device: kinds, areas, and mesh-light leaf orders
int rtrh_lights(void *hs, int32_t *kind, int32_t *prim, float *area, int32_t *tri_begin,
                int32_t *tris, int32_t *n_tris) {
  HostScene *h = (HostScene *)hs;
  int n = -1;
  guarded([&] {
    const FlatScene &f = h->scene->flatten();
    n = (int)f.light_kind.size();
    for (int i = 0; i < n; i++) {
      if (kind) kind[i] = f.light_kind[i];
      if (prim) prim[i] = f.light_prim[i];
      if (area) area[i] = f.light_area[i];
    }
    if (tri_begin) memcpy(tri_begin, f.light_tri_begin.data(), 4 * f.light_tri_begin.size());
    if (tris) memcpy(tris, f.light_tris.data(), 4 * f.light_tris.size());
    if (n_tris) *n_tris = (int32_t)f.light_tris.size();
  });
  return n;
}

int rtrh_set_device(void *hs, int device) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] { h->scene->setDevice(device); });
}

// the rtr_scene* behind this scene (uploads on first use); NULL on failure
rtr_scene *rtrh_device_scene(void *hs) {
  HostScene *h = (HostScene *)hs;
  rtr_scene *s = nullptr;
  guarded([&] { s = h->scene->deviceScene(); });
  return s;
}

// the flattened scene as the C ABI takes it (valid until the scene changes); NULL on failure
const rtr_scene_desc *rtrh_scene_desc(void *hs) {
  HostScene *h = (HostScene *)hs;
  const rtr_scene_desc *out = nullptr;
  guarded([&] {
    h->desc = h->scene->flatten().desc();
    out = &h->desc;
  });
  return out;
}

int rtrh_material_id(void *hs, int material) {
  // id of the host material in the flattened table (-1 if no primitive uses it)
  HostScene *h = (HostScene *)hs;
  int id = -1;
  guarded([&] {
    const FlatScene &f = h->scene->flatten();
    const Material *m = h->mats.at(material);
    for (size_t p = 0; p < f.prim_object.size(); p++)
      if (f.prim_object[p]->getMaterial() == m) {
        id = f.prim_material[p];
        break;
      }
  });
  return id;
}

int rtrh_intersect(void *hs, const float *org, const float *dir, int64_t n, int32_t *prim, float *t,
                   float *pos, float *nrm) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] { h->scene->intersectBatch(org, dir, (size_t)n, prim, t, pos, nrm); });
}

int rtrh_set_node_box(void *hs, int flat_index, const float *lo, const float *hi) {
  HostScene *h = (HostScene *)hs;
  bool found = false;
  int rc = guarded([&] { found = h->scene->setNodeBound(flat_index, Bound3(v3(lo), v3(hi))); });
  return rc ? rc : (found ? 0 : -1);
}

int rtrh_check_connectable(void *hs, const float *a, const float *b, int64_t n, uint8_t *connected) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] { h->scene->checkConnectableBatch(a, b, (size_t)n, connected); });
}

// Render::createPhotonMap through the C++ mirror; arrays hold n rows; returns rows filled
int64_t rtrh_create_photon_map(void *hs, uint32_t seed, uint32_t n, float *pos, float *wi, float *power) {
  HostScene *h = (HostScene *)hs;
  int64_t filled = 0;
  int rc = guarded([&] {
    h->render.options.seed = seed;
    std::vector<Photon *> map;
    h->render.createPhotonMap(*h->scene, &map, n);
    for (Photon *ph : map) {
      if (!ph) continue;
      const Vector3f v[3] = {ph->pos, ph->wi, ph->power};
      float *dst[3] = {pos, wi, power};
      for (int a = 0; a < 3; a++) {
        dst[a][3 * filled] = v[a].x;
        dst[a][3 * filled + 1] = v[a].y;
        dst[a][3 * filled + 2] = v[a].z;
      }
      filled++;
      delete ph;
    }
  });
  return rc ? -1 : filled;
}

// RenderOptions::devices for the next rtrh_render calls (n = 0: back to the scene's own device)
int rtrh_set_render_devices(void *hs, const int *devices, int n) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] { h->render.options.devices.assign(devices, devices + (n > 0 ? n : 0)); });
}

// Render::renderPathTracing; file may be NULL/"" (no PPM).  rgb gets the buffer.
int rtrh_render(void *hs, unsigned W, unsigned H, unsigned spp, const char *file,
                const rtr_render_params *opt, float *rgb, rtr_stats *stats) {
  HostScene *h = (HostScene *)hs;
  return guarded([&] {
    if (opt) {
      h->render.options.seed = opt->seed;
      h->render.options.max_depth = opt->max_depth;
      h->render.options.rr_prob = opt->rr_prob;
      h->render.options.chunk_spp = opt->chunk_spp;
      h->render.options.pool_size = opt->pool_size;
      h->render.options.part_index = opt->part_index;
      h->render.options.part_count = opt->part_count > 0 ? opt->part_count : 1;
      h->render.options.tile_cols = opt->tile_cols;
      h->render.options.profile = opt->profile != 0;
      h->render.options.mode = opt->mode;
      h->render.options.streams = opt->streams;
    }
    if (!h->render.options.devices.empty()) h->render.options.part_count = 1;
    h->render.renderPathTracing(*h->scene, W, H, spp, file ? std::string(file) : std::string());
    const std::vector<Vector3f> &b = h->render.getBuffer();
    if (rgb) memcpy(rgb, b.data(), b.size() * sizeof(Vector3f));
    if (stats) *stats = h->render.getStats();
  });
}

// Render::saveFile on a caller's buffer of height * width pixels (three floats each)
int rtrh_save_file(const float *rgb, unsigned height, unsigned width, const char *file) {
  return guarded([&] {
    if (!rgb || !file) throw std::invalid_argument("rtrh_save_file: null argument");
    std::vector<Vector3f> px((size_t)height * width);
    memcpy(px.data(), rgb, px.size() * sizeof(Vector3f));
    Render::saveFile(px, std::string(file), height, width);
  });
}

} /* extern "C" */
