/*
 * ref_harness.cpp -- drives the REFERENCE implementation itself.
 *
 * TEST INFRASTRUCTURE ONLY.  This translation unit #includes the reference's
 * own, unmodified source files from the read-only checkout (path given by the
 * Makefile; nothing is copied into this repository) and exposes a small C
 * interface over them so the tests can (a) validate oracle/pt_oracle.cpp and
 * (b) generate the golden vectors under tests/golden/.
 *
 * All reference .cpp files are compiled as ONE translation unit on purpose:
 * include/Global.h defines `static std::mt19937 rng` per translation unit, so
 * a single unit gives a single generator that this harness can re-seed, which
 * makes the reference's random draws reproducible.
 *
 * Only built where the reference checkout exists; the resulting library goes
 * to oracle/_ref/ (git-ignored).
 */
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <queue>
#include <random>
#include <stack>
#include <thread>
#include <vector>
#include <cstring>
#include <cstdint>

/* reach Render::pathTracing / Render::buffer / BVH::root / TrangleMesh::trangles */
#define private public
#define protected public
#include "Sphere.h"
#include "Render.h"
#include "Trangle.h"
#include "Bound3.cpp"
#include "BVH.cpp"
#include "Material.cpp"
#include "Scene.cpp"
#include "Sphere.cpp"
#include "Trangle.cpp"
#include "Render.cpp"
#undef private
#undef protected

namespace {

struct RefScene {
  Scene *scene = nullptr; /* never destroyed: Scene::~Scene double-frees its BVH */
  Camera *camera = nullptr;
  std::vector<Material *> mats;
  std::vector<Object *> objects;     /* add order */
  std::vector<int> kinds;            /* 0 sphere, 1 mesh */
  std::map<Object *, int> prim_of;   /* Sphere* / Trangle* -> unified prim id */
  std::map<Object *, int> tri_index; /* Trangle* -> position of its index triple in the mesh */
  std::map<Object *, int> object_of; /* top-level Object* -> object id */
  Render render;
  bool built = false;
};

void assign_ids(RefScene *r) {
  int p = 0;
  r->prim_of.clear();
  for (size_t o = 0; o < r->objects.size(); o++) {
    r->object_of[r->objects[o]] = (int)o;
    if (r->kinds[o] == 0) {
      r->prim_of[r->objects[o]] = p++;
    } else {
      TrangleMesh *m = (TrangleMesh *)r->objects[o];
      /* BVH::selfBuildBVH sorts TrangleMesh::trangles in place (src/BVH.cpp:42-55 via
       * include/Trangle.h:93), so ids come from the recorded construction index. */
      for (size_t k = 0; k < m->trangles.size(); k++)
        r->prim_of[m->trangles[k]] = p + r->tri_index[m->trangles[k]];
      p += (int)m->trangles.size();
    }
  }
}

struct Flat {
  std::vector<float> lo, hi;
  std::vector<int32_t> child, prim;
};

int flat_node(RefScene *r, BVHNode *n, bool top, Flat &f) {
  if (top && n->object != nullptr && r->kinds[r->object_of[n->object]] == 1) {
    TrangleMesh *m = (TrangleMesh *)n->object;
    return flat_node(r, m->bvh->root, false, f);
  }
  int id = (int)f.prim.size();
  f.lo.insert(f.lo.end(), {n->bound.min.x, n->bound.min.y, n->bound.min.z});
  f.hi.insert(f.hi.end(), {n->bound.max.x, n->bound.max.y, n->bound.max.z});
  f.child.push_back(-1);
  f.child.push_back(-1);
  f.prim.push_back(-1);
  if (n->object != nullptr) {
    f.prim[id] = r->prim_of[n->object];
  } else {
    int l = flat_node(r, n->left, top, f);
    int rr = flat_node(r, n->right, top, f);
    f.child[2 * id] = l;
    f.child[2 * id + 1] = rr;
  }
  return id;
}

/* the walk of flat_node, looking for flattened node `want`: overwrites the reference's own
 * BVHNode::bound (the top-level leaf of a mesh and the mesh's root are one flattened node) */
bool set_bound(RefScene *r, BVHNode *n, bool top, int &next, int want, const Bound3 &b) {
  if (top && n->object != nullptr && r->kinds[r->object_of[n->object]] == 1) {
    TrangleMesh *m = (TrangleMesh *)n->object;
    if (next == want) n->bound = b;
    return set_bound(r, m->bvh->root, false, next, want, b);
  }
  if (next++ == want) {
    n->bound = b;
    return true;
  }
  if (n->object != nullptr) return false;
  return set_bound(r, n->left, top, next, want, b) || set_bound(r, n->right, top, next, want, b);
}

void top_leaves(RefScene *r, BVHNode *n, std::vector<int32_t> &out) {
  if (n->object != nullptr) {
    out.push_back(r->object_of[n->object]);
  } else {
    top_leaves(r, n->left, out);
    top_leaves(r, n->right, out);
  }
}

inline Vector3f v3(const float *p) { return Vector3f(p[0], p[1], p[2]); }

} // namespace

extern "C" {

void *ref_scene_new(const float pos[3], const float front[3], const float up[3], float fov) {
  RefScene *r = new RefScene();
  r->camera = new Camera(v3(pos), v3(front), v3(up), fov);
  r->scene = new Scene(r->camera);
  return r;
}

int ref_add_material(void *h, int kind, const float e[3], const float kd[3], const float ks[3],
                     float ior, float roughness) {
  RefScene *r = (RefScene *)h;
  r->mats.push_back(new Material((MaterialType)kind, v3(e), v3(kd), v3(ks), ior, roughness));
  return (int)r->mats.size() - 1;
}

int ref_add_sphere(void *h, int material, const float c[3], float radius) {
  RefScene *r = (RefScene *)h;
  Sphere *s = new Sphere(r->mats[material], v3(c), radius);
  r->objects.push_back((Object *)s);
  r->kinds.push_back(0);
  r->scene->addObject((Object *)s);
  return (int)r->objects.size() - 1;
}

int ref_add_mesh(void *h, int material, const float *positions, int n_pos, const uint32_t *idx,
                 int n_idx) {
  RefScene *r = (RefScene *)h;
  std::vector<Vector3f> pos(n_pos);
  for (int i = 0; i < n_pos; i++) pos[i] = v3(positions + 3 * i);
  std::vector<unsigned int> ind(idx, idx + n_idx);
  TrangleMesh *m = new TrangleMesh(r->mats[material], &pos, &ind);
  /* recover each Trangle's construction index from its vertices (the ctor has
   * already re-ordered the vector); identical triangles take indices in turn */
  std::map<std::vector<float>, std::vector<int>> by_vertices;
  for (int i = n_idx / 3 - 1; i >= 0; i--) {
    std::vector<float> key;
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) key.push_back(positions[3 * idx[3 * i + k] + a]);
    by_vertices[key].push_back(i);
  }
  for (size_t k = 0; k < m->trangles.size(); k++) {
    Trangle *t = (Trangle *)m->trangles[k];
    std::vector<float> key = {t->p0.x, t->p0.y, t->p0.z, t->p1.x, t->p1.y, t->p1.z,
                              t->p2.x, t->p2.y, t->p2.z};
    std::vector<int> &slot = by_vertices[key];
    r->tri_index[m->trangles[k]] = slot.back();
    slot.pop_back();
  }
  r->objects.push_back((Object *)m);
  r->kinds.push_back(1);
  r->scene->addObject((Object *)m);
  return (int)r->objects.size() - 1;
}

void ref_build(void *h) {
  RefScene *r = (RefScene *)h;
  r->scene->rebuildBVH();
  assign_ids(r);
  r->built = true;
}

int ref_flatten(void *h, float *bmin3, float *bmax3, int32_t *child2, int32_t *prim) {
  RefScene *r = (RefScene *)h;
  Flat f;
  flat_node(r, r->scene->bvh->root, true, f);
  int n = (int)f.prim.size();
  if (bmin3) memcpy(bmin3, f.lo.data(), sizeof(float) * 3 * n);
  if (bmax3) memcpy(bmax3, f.hi.data(), sizeof(float) * 3 * n);
  if (child2) memcpy(child2, f.child.data(), sizeof(int32_t) * 2 * n);
  if (prim) memcpy(prim, f.prim.data(), sizeof(int32_t) * n);
  return n;
}

int ref_top_leaf_order(void *h, int32_t *object_ids) {
  RefScene *r = (RefScene *)h;
  std::vector<int32_t> v;
  top_leaves(r, r->scene->bvh->root, v);
  if (object_ids) memcpy(object_ids, v.data(), sizeof(int32_t) * v.size());
  return (int)v.size();
}

float ref_object_area(void *h, int object) {
  RefScene *r = (RefScene *)h;
  return r->objects[object]->getArea();
}

void ref_intersect(void *h, const float *org3, const float *dir3, int64_t n, int32_t *prim,
                   float *t, float *pos3, float *nrm3) {
  RefScene *r = (RefScene *)h;
  for (int64_t i = 0; i < n; i++) {
    Ray ray(v3(org3 + 3 * i), v3(dir3 + 3 * i));
    Intersection it = r->scene->intersect(ray);
    prim[i] = it.hit ? r->prim_of[it.object] : -1;
    t[i] = it.distance;
    if (pos3) { pos3[3 * i] = it.postion.x; pos3[3 * i + 1] = it.postion.y; pos3[3 * i + 2] = it.postion.z; }
    if (nrm3) { nrm3[3 * i] = it.normal.x; nrm3[3 * i + 1] = it.normal.y; nrm3[3 * i + 2] = it.normal.z; }
  }
}

void ref_check_connectable(void *h, const float *a3, const float *b3, int64_t n, uint8_t *connected) {
  RefScene *r = (RefScene *)h;
  for (int64_t i = 0; i < n; i++)
    connected[i] = r->scene->checkConnectable(v3(a3 + 3 * i), v3(b3 + 3 * i)) ? 1 : 0;
}

int ref_slab(const float bmin[3], const float bmax[3], const float org[3], const float dir[3]) {
  Bound3 b(v3(bmin), v3(bmax));
  Ray ray(v3(org), v3(dir));
  return b.intersect(ray) ? 1 : 0;
}

void ref_material_eval(void *h, int material, const float *wi3, const float *wo3, const float *n3,
                       int64_t count, float *out3) {
  RefScene *r = (RefScene *)h;
  for (int64_t i = 0; i < count; i++) {
    Vector3f v = r->mats[material]->eval(v3(wi3 + 3 * i), v3(wo3 + 3 * i), v3(n3 + 3 * i));
    out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
  }
}

void ref_material_pdf(void *h, int material, const float *wi3, const float *wo3, const float *n3,
                      int64_t count, float *out) {
  RefScene *r = (RefScene *)h;
  for (int64_t i = 0; i < count; i++)
    out[i] = r->mats[material]->pdf(v3(wi3 + 3 * i), v3(wo3 + 3 * i), v3(n3 + 3 * i));
}

/* the one generator of this translation unit */
void ref_rng_seed(uint32_t seed) {
  rng.seed(seed);
  dist.reset();
}
void ref_rng_draw(int n, float *out) {
  for (int i = 0; i < n; i++) out[i] = getRandomFloat();
}

/* Material::sample after seeding; each call i uses seed+i so the caller can
 * replay the identical draws from ref_rng_seed/ref_rng_draw. */
void ref_material_sample(void *h, int material, const float *wi3, const float *n3, uint32_t seed,
                         int64_t count, float *wo3) {
  RefScene *r = (RefScene *)h;
  for (int64_t i = 0; i < count; i++) {
    ref_rng_seed(seed + (uint32_t)i);
    Vector3f v = r->mats[material]->sample(v3(wi3 + 3 * i), v3(n3 + 3 * i));
    wo3[3 * i] = v.x; wo3[3 * i + 1] = v.y; wo3[3 * i + 2] = v.z;
  }
}

void ref_sample_light(void *h, uint32_t seed, float pos[3], float nrm[3], float *pdf,
                      int32_t *prim) {
  RefScene *r = (RefScene *)h;
  ref_rng_seed(seed);
  Intersection it;
  float p = -1.0f;
  r->scene->samplingLight(it, p);
  pos[0] = it.postion.x; pos[1] = it.postion.y; pos[2] = it.postion.z;
  nrm[0] = it.normal.x; nrm[1] = it.normal.y; nrm[2] = it.normal.z;
  *pdf = p;
  *prim = it.object ? r->prim_of[it.object] : -1;
}

/* Render::pathTracing for one ray, single-threaded, generator seeded first */
void ref_path_trace(void *h, const float org[3], const float dir[3], uint32_t seed, float rgb[3]) {
  RefScene *r = (RefScene *)h;
  ref_rng_seed(seed);
  Ray ray(v3(org), v3(dir));
  Vector3f c = r->render.pathTracing(*r->scene, ray);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* Overwrites the box of flattened node `flat_index` in the reference's tree. */
int ref_set_node_box(void *h, int flat_index, const float lo[3], const float hi[3]) {
  RefScene *r = (RefScene *)h;
  int next = 0;
  Bound3 b(v3(lo), v3(hi));
  return set_bound(r, r->scene->bvh->root, true, next, flat_index, b) ? 0 : -1;
}

/* Render::createPhotonMap after seeding.  photon_num must be below 16: the reference then
 * runs its thread function once over [0, photon_num) (src/Render.cpp:467-470), so the draws
 * are reproducible; from 16 on, 16 threads race on the one generator.  Returns the number
 * of map entries that were filled. */
int ref_create_photon_map(void *h, uint32_t seed, unsigned photon_num, float *pos3, float *wi3,
                          float *power3) {
  RefScene *r = (RefScene *)h;
  if (photon_num >= 16) return -1;
  ref_rng_seed(seed);
  std::vector<Photon *> map(photon_num, nullptr);
  r->render.createPhotonMap(*r->scene, &map, photon_num);
  int k = 0;
  for (unsigned i = 0; i < photon_num; i++) {
    if (!map[i]) continue;
    const Vector3f v[3] = {map[i]->pos, map[i]->wi, map[i]->power};
    float *dst[3] = {pos3, wi3, power3};
    for (int a = 0; a < 3; a++) { dst[a][3 * k] = v[a].x; dst[a][3 * k + 1] = v[a].y; dst[a][3 * k + 2] = v[a].z; }
    k++;
  }
  return k;
}

/* Render::renderPathTracing (16 threads, unsynchronised generator, writes
 * <file>.ppm); copies the private float buffer out.  Returns 0 on success. */
int ref_render(void *h, unsigned W, unsigned H, unsigned spp, const char *file, float *rgb) {
  RefScene *r = (RefScene *)h;
  {
    FILE *probe = fopen((std::string(file) + ".ppm").c_str(), "wb");
    if (!probe) return -1; /* saveFile would dereference a null FILE* */
    fclose(probe);
  }
  r->render.renderPathTracing(*r->scene, W, H, spp, std::string(file));
  memcpy(rgb, r->render.buffer->data(), sizeof(float) * 3 * (size_t)W * H);
  return 0;
}

/* Render::saveFile (src/Render.cpp:3-21) on a caller's float buffer of height * width pixels:
 * writes <file>.ppm.  Returns 0 on success. */
int ref_save_file(const float *rgb, unsigned height, unsigned width, const char *file) {
  {
    FILE *probe = fopen((std::string(file) + ".ppm").c_str(), "wb");
    if (!probe) return -1; /* saveFile would dereference a null FILE* */
    fclose(probe);
  }
  Render render;
  render.buffer = new std::vector<Vector3f>((size_t)height * width);
  for (size_t i = 0; i < (size_t)height * width; i++) (*render.buffer)[i] = v3(rgb + 3 * i);
  render.saveFile(std::string(file), height, width);
  delete render.buffer;
  render.buffer = nullptr;
  return 0;
}

} /* extern "C" */
