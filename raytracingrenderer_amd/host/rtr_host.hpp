// Host-side mirror of the reference's scene API for the path-tracing path.
//
// A program written against the reference (src/main.cpp:7-90) builds Material,
// Sphere, TrangleMesh, Camera and Scene objects, calls Scene::rebuildBVH and then
// Render::renderPathTracing.  The classes below keep those names, constructor
// signatures and argument meanings, so such a program compiles against this
// header unchanged; what differs is where the work happens.  Nothing here
// intersects or shades on the CPU: the Scene is flattened into the arrays of
// include/rtr.h and every query or render goes through that C ABI to the HIP
// kernels.  If the device library cannot run, calls throw std::runtime_error
// with the library's message -- there is no CPU fallback.
//
// The tree built here must have the reference's exact topology and bounds, so
// the build (longest axis, std::sort on box centres, split at size/2) follows
// src/BVH.cpp:19-64 step by step, including the in-place sort of the caller's
// vector.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtr.h"

#define MY_PI 3.1415926  // include/Global.h:4

// ---- include/Vector.h ----
class Vector2f {
 public:
  float x, y;
  Vector2f() : x(0), y(0) {}
  Vector2f(float v) : x(v), y(v) {}
  Vector2f(float _x, float _y) : x(_x), y(_y) {}
};

class Vector3f {
 public:
  float x, y, z;
  Vector3f() : x(0), y(0), z(0) {}
  Vector3f(float v) : x(v), y(v), z(v) {}
  Vector3f(float _x, float _y, float _z) : x(_x), y(_y), z(_z) {}
  Vector3f operator+(const Vector3f &o) const { return Vector3f(x + o.x, y + o.y, z + o.z); }
  Vector3f operator-(const Vector3f &o) const { return Vector3f(x - o.x, y - o.y, z - o.z); }
  Vector3f operator-() const { return Vector3f(-x, -y, -z); }
  Vector3f operator*(float n) const { return Vector3f(x * n, y * n, z * n); }
  Vector3f operator*(const Vector3f &o) const { return Vector3f(x * o.x, y * o.y, z * o.z); }
  Vector3f operator/(float n) const { return Vector3f(x / n, y / n, z / n); }
  friend Vector3f operator*(float n, const Vector3f &v) { return Vector3f(n * v.x, n * v.y, n * v.z); }
  float getLen() const;
  float norm() const { return x * x + y * y + z * z; }  // squared length, include/Vector.h:82
  Vector3f normalize() const;
};

namespace vec {
float dotProduct(const Vector3f &a, const Vector3f &b);
Vector3f crossProduct(const Vector3f &a, const Vector3f &b);
}  // namespace vec

// ---- include/Ray.h, include/Intersection.h ----
struct Ray {
  Vector3f startPos, dir;
  Ray(Vector3f s, Vector3f d) : startPos(s), dir(d) {}
};

class Object;
struct Intersection {
  bool hit = false;
  float distance = 0;
  Vector3f postion;  // (sic) the reference's spelling
  Vector3f normal;
  Vector2f textureCoord;
  Object *object = nullptr;
};

// ---- include/Bound3.h ----
class Bound3 {
 public:
  Vector3f min, max;
  Bound3();  // empty box: min = +FLT_MAX, max = lowest
  Bound3(Vector3f lo, Vector3f hi) : min(lo), max(hi) {}
  Bound3 &merge(const Bound3 &b);
  Vector3f getCenterPos() const;
};

// ---- include/Material.h ----
enum MaterialType { DIFFUSE, SPECULAR, MICROFACET };

class Material {
  MaterialType type;
  Vector3f emission, kd, ks;
  float ior, specularExponent, roughness;

 public:
  Material(MaterialType _type, Vector3f _emission, Vector3f _kd = Vector3f(0),
           Vector3f _ks = Vector3f(0), float _ior = 1e-5, float _roughness = 1e-5,
           float _specularExponent = 32);
  MaterialType getMaterialType() const { return type; }
  bool hasEmission() const;
  Vector3f getEmission() const { return emission; }
  Vector3f getKd() const { return kd; }
  Vector3f getKs() const { return ks; }
  float getIor() const { return ior; }
  float getRoughness() const { return roughness; }
  float getSpecularExponent() const { return specularExponent; }
};

// ---- include/Object.h ----
class Object {
 protected:
  Material *material;
  Bound3 bound;
  float area = 0;

 public:
  enum Kind { SPHERE, TRIANGLE, MESH };
  explicit Object(Material *m) : material(m) {}
  virtual ~Object() {}
  const Material *getMaterial() const { return material; }
  Bound3 getBound() const { return bound; }
  virtual float getArea() const { return area; }
  virtual Kind kind() const = 0;
};

// ---- include/BVH.h ----
struct BVHNode {
  Bound3 bound;
  Object *object = nullptr;
  BVHNode *left = nullptr, *right = nullptr;
};

class BVH {
  BVHNode *root = nullptr;
  BVHNode *arena = nullptr;  // one block holding the nodes of a tree that came from the device as arrays
  static BVHNode *build(std::vector<Object *> &objs);
  BVHNode *buildOnDevice(std::vector<Object *> &objs, bool finite_only);
  static void destroy(BVHNode *n);
  void clear();

 public:
  // Where trees are built.  Auto: on the device (rtr_build_bvh) from deviceMin objects on when a
  // GPU is present, else by std::sort on the host; both give the reference's tree bit for bit.
  // The environment variable RTR_TREE_BUILDER = host | device overrides the setting.
  enum Builder { Auto, Host, Device };
  static Builder builder;
  static size_t deviceMin;
  static int buildDevice;  // negative: the calling thread's current HIP device
  BVH() {}
  ~BVH() { clear(); }
  BVH(const BVH &) = delete;
  BVH &operator=(const BVH &) = delete;
  // sorts objs in place like the reference does (src/BVH.cpp:42-55)
  void buildBVH(std::vector<Object *> &objs);
  Bound3 getBound() const { return root->bound; }
  const BVHNode *getRoot() const { return root; }
  // non-null: the tree's nodes are one block in depth-first order, left child first (root at [0])
  const BVHNode *preorderBlock() const { return arena; }
};

// ---- include/Sphere.h ----
class Sphere : public Object {
  Vector3f center;
  float radius;

 public:
  Sphere(Material *_material, Vector3f _center, float _radius);
  Kind kind() const override { return SPHERE; }
  Vector3f getCenter() const { return center; }
  float getRadius() const { return radius; }
};

// ---- include/Trangle.h ----
class Trangle : public Object {
  Vector3f p0, p1, p2, normal;
  Vector2f uv0, uv1, uv2;

 public:
  int indexInMesh = 0;  // position of this triangle's index triple in its mesh
  Trangle(Material *_material, Vector3f _p0, Vector3f _p1, Vector3f _p2, Vector3f _normal,
          Vector2f _uv0, Vector2f _uv1, Vector2f _uv2);
  Kind kind() const override { return TRIANGLE; }
  const Vector3f &P0() const { return p0; }
  const Vector3f &P1() const { return p1; }
  const Vector3f &P2() const { return p2; }
  const Vector3f &getNormal() const { return normal; }
};

class TrangleMesh : public Object {
  std::vector<Object *> trangles;  // re-ordered by the BVH build, like the reference's
  Trangle *block = nullptr;        // the triangles themselves, one allocation for the whole mesh
  BVH *bvh;

 public:
  TrangleMesh(Material *_material, std::vector<Vector3f> *_positions,
              std::vector<unsigned int> *_posIndices, std::vector<Vector3f> *_normals = nullptr,
              std::vector<Vector2f> *_textureCoords = nullptr);
  ~TrangleMesh() override;
  TrangleMesh(const TrangleMesh &) = delete;
  TrangleMesh &operator=(const TrangleMesh &) = delete;
  Kind kind() const override { return MESH; }
  size_t numTrangles() const { return trangles.size(); }
  const std::vector<Object *> &getTrangles() const { return trangles; }
  const BVH *getBVH() const { return bvh; }
};

// ---- include/Scene.h ----
struct Camera {
  Vector3f position, frontDir, upDir;
  float fov;
  Camera(Vector3f p, Vector3f f, Vector3f u, float _fov) : position(p), frontDir(f), upDir(u), fov(_fov) {}
};

// std::allocator whose resize() leaves trivial elements uninitialised: the large arrays below are
// sized first and then filled by several threads, and zeroing them first (on one thread) costs as
// much as filling them.
template <class T>
struct NoInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    typedef NoInitAllocator<U> other;
  };
  NoInitAllocator() = default;
  template <class U>
  NoInitAllocator(const NoInitAllocator<U> &) {}
  template <class U>
  void construct(U *p) {
    ::new (static_cast<void *>(p)) U;
  }
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
  }
};
template <class T>
using BulkVector = std::vector<T, NoInitAllocator<T>>;

// The arrays of rtr_scene_desc, owned.
struct FlatScene {
  BulkVector<float> node_bmin[3], node_bmax[3];
  BulkVector<int32_t> node_left, node_right, node_prim;
  BulkVector<int32_t> prim_kind, prim_index, prim_material;
  BulkVector<float> tri_p0[3], tri_p1[3], tri_p2[3], tri_n[3], tri_area;
  std::vector<float> sph_c[3], sph_r, sph_area;
  std::vector<int32_t> mat_kind;
  std::vector<float> mat_emission[3], mat_kd[3], mat_ks[3], mat_ior, mat_roughness;
  std::vector<int32_t> light_kind, light_prim, light_tri_begin, light_tris;
  std::vector<float> light_area;
  BulkVector<Object *> prim_object;  // prim id -> Sphere* / Trangle*
  rtr_scene_desc desc() const;
};

class Scene {
  Camera *camera;
  std::vector<Object *> objects;   // re-ordered by rebuildBVH, like the reference's
  std::vector<Object *> addOrder;  // objects in addObject order: defines prim ids
  std::vector<Object *> lights;
  BVH *bvh;
  rtr_scene *device = nullptr;
  std::vector<rtr_scene *> replicas;  // deviceScenes(): one upload per entry of its list
  std::vector<int> replicaDevices;
  FlatScene flat;
  int deviceIndex = 0;
  bool flatValid = false;
  void dropReplicas();

 public:
  explicit Scene(Camera *_camera);
  ~Scene();
  Scene(const Scene &) = delete;
  Scene &operator=(const Scene &) = delete;

  const Camera *getCamera() const { return camera; }
  void addObject(Object *obj);
  void rebuildBVH();
  // Scene::intersect of the reference, answered by the device (one-ray batch)
  Intersection intersect(const Ray &ray);
  // many rays at once: prim ids (-1 = miss) and distances; pos/nrm optional (n*3)
  void intersectBatch(const float *org_xyz, const float *dir_xyz, size_t n, int32_t *prim_id,
                      float *t, float *pos_xyz = nullptr, float *nrm_xyz = nullptr);
  // include/Scene.h:49-56 (ESP is ignored there, and here), and many pairs at once
  bool checkConnectable(Vector3f a, Vector3f b, float ESP = 1e-3);
  void checkConnectableBatch(const float *a_xyz, const float *b_xyz, size_t n, uint8_t *connected);

  void setDevice(int index);
  // Overwrites the box of node flatIndex (numbering of flatten()): for trees built elsewhere,
  // whose boxes need not contain their children's.  Returns false if there is no such node.
  bool setNodeBound(int flatIndex, const Bound3 &b);
  const FlatScene &flatten();  // (re)built after rebuildBVH
  rtr_scene *deviceScene();    // uploaded on first use; throws if no device
  // The scene on each device of the list (uploaded on first use, kept until the list or the
  // scene changes).  A device may appear more than once: every entry is a replica of its own.
  const std::vector<rtr_scene *> &deviceScenes(const std::vector<int> &devices);
  const std::vector<Object *> &getLights() const { return lights; }
};

// ---- include/Render.h ----
enum class RenderType { PathTracing };

struct RenderOptions {  // what the reference hard-codes, see rtr_render_params
  uint32_t seed = 0;
  int max_depth = 0;
  float rr_prob = 0;
  int chunk_spp = 0, pool_size = 0;
  int part_index = 0, part_count = 1, tile_cols = 0;  // tile_cols 0: 8, or balanced tiles with `devices`
  bool profile = false;
  int mode = 0, streams = 0;  // rtr_render_params::mode / ::streams
  // GPUs to render on, e.g. {0, 1, 2, 3, 4, 5, 6, 7}: the image is cut into interleaved column
  // tiles, one partition per entry, rendered concurrently and gathered on the first device
  // (rtr_render_pt_multi).  Empty: the scene's own device (Scene::setDevice).  part_index /
  // part_count must be left alone when this is set.
  std::vector<int> devices;
};

// ---- include/Photon.h ----
struct Photon {
  Vector3f pos;
  Vector3f wi;
  Vector3f power;
};

class Render {
  std::vector<Vector3f> *buffer = nullptr;
  rtr_stats lastStats{};

 public:
  RenderOptions options;
  Render() {}
  ~Render() { delete buffer; }
  // src/Render.cpp:632-694: renders into the internal buffer (index i*height+j)
  // and writes <fileName>.ppm unless fileName is empty.
  void renderPathTracing(Scene &scene, const unsigned int &imageWidth,
                         const unsigned int &imageHeight, const unsigned int &spp,
                         std::string fileName);
  void render(Scene &scene, const unsigned int &imageWidth, const unsigned int &imageHeight,
              const unsigned int &spp, std::string fileName, RenderType type);
  // Render::createPhotonMap, src/Render.cpp:405-481: fills photonMap (resized to
  // photonNum) with newly allocated photons the caller owns, like the reference; entries
  // stay null when the scene has no light.  Draws come from options.seed (see
  // rtr_create_photon_map).  Public here; the reference keeps it private to its
  // photon-mapping integrator.
  void createPhotonMap(Scene &scene, std::vector<Photon *> *photonMap, unsigned int photonNum);
  const std::vector<Vector3f> &getBuffer() const { return *buffer; }
  const rtr_stats &getStats() const { return lastStats; }
  // Render::saveFile, src/Render.cpp:3-21 (public here so tools can reuse it)
  static void saveFile(const std::vector<Vector3f> &pixels, const std::string &fileName,
                       unsigned int height, unsigned int width);
};
