/*
 * pt_oracle.h -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library may include, link
 * or call this.  It exists so the tests (and the CPU leg of the benchmark) can
 * check the HIP kernels against an independent scalar implementation of the
 * reference algorithm.  Every function cites the reference file:line it
 * restates (paths relative to the reference checkout).
 *
 * Parity pin: the restatement is checked against the reference itself, built
 * from its own sources by oracle/Makefile into oracle/_ref/libref.so, both
 * live (tests/test_oracle_vs_ref.py) and through the committed vectors under
 * tests/golden/ that were generated from that build.
 *
 * Everything is plain C at the interface so ctypes can drive it.  The
 * implementation is C++ only because the reference's BVH build order is
 * defined by libstdc++'s std::sort on tied keys.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* material kinds, include/Material.h:6-11 */
enum { ORC_DIFFUSE = 0, ORC_SPECULAR = 1, ORC_MICROFACET = 2 };

/* How transcendental calls are evaluated.
 *   ORC_MATH_LIBM : cosf/sinf/powf from the host libm, as the reference calls
 *                   them -- used when pinning against oracle/_ref.
 *   ORC_MATH_PORTABLE : a fixed sequence of IEEE fp32 add/mul (no FMA) for
 *                   sin/cos and t*t for powf(t,2) -- reproducible bit for bit
 *                   on any IEEE machine, used when checking the GPU. */
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };

/* uniform [0,1) source for the reference-order integrator */
typedef float (*orc_rng_fn)(void *user);

/* ---- scene construction (src/main.cpp:7-84 call pattern) ---- */
orc_scene *orc_scene_new(const float cam_pos[3], const float cam_front[3],
                         const float cam_up[3], float fov);
void orc_scene_free(orc_scene *s);
/* Material ctor clamps, include/Material.h:53-61.  Returns material id. */
int orc_add_material(orc_scene *s, int kind, const float emission[3],
                     const float kd[3], const float ks[3], float ior,
                     float roughness);
/* Sphere ctor, include/Sphere.h:16-21.  Returns object id. */
int orc_add_sphere(orc_scene *s, int material, const float center[3],
                   float radius);
/* TrangleMesh ctor (positions + index triples, no normals/uvs),
 * include/Trangle.h:60-95.  Returns object id. */
int orc_add_mesh(orc_scene *s, int material, const float *positions,
                 int n_positions, const uint32_t *indices, int n_indices);
/* Scene::rebuildBVH, include/Scene.h:44 -> src/BVH.cpp:19-64 */
void orc_build(orc_scene *s);
/* the scene hard-coded in src/main.cpp:7-84 */
orc_scene *orc_default_scene(void);
void orc_set_math(orc_scene *s, int mode);

/* ---- introspection ---- */
int orc_num_prims(const orc_scene *s);
int orc_num_objects(const orc_scene *s);
int orc_num_lights(const orc_scene *s);
/* Spliced tree (mesh leaves replaced by the mesh's own tree), DFS pre-order.
 * node_child: left,right node index (-1,-1 for a leaf); node_prim: prim id or
 * -1.  Call with NULL arrays to get the node count. */
int orc_flatten(const orc_scene *s, float *bmin3, float *bmax3,
                int32_t *node_child2, int32_t *node_prim);
/* leaf prim ids in left-to-right DFS order of the top-level tree (object ids)*/
int orc_top_leaf_order(const orc_scene *s, int32_t *object_ids);
/* Overwrites the box of node `flat_index` (numbering of orc_flatten), so that trees whose
 * boxes are NOT nested can be checked: the reference tests whatever box a node carries
 * (src/BVH.cpp:66-80), a child is only reached through its ancestors' boxes. */
int orc_set_node_box(orc_scene *s, int flat_index, const float lo[3], const float hi[3]);
float orc_object_area(const orc_scene *s, int object);

/* ---- closest hit: Scene::intersect, include/Scene.h:46 ---- */
/* prim: unified primitive id (objects in add order; a mesh contributes its
 * triangles in index order), -1 on miss.  pos/nrm may be NULL. */
void orc_intersect(const orc_scene *s, const float *org3, const float *dir3,
                   int64_t n, int32_t *prim, float *t, float *pos3,
                   float *nrm3);
/* Bound3::intersect on an explicit box, src/Bound3.cpp:3-22 */
int orc_slab(const float bmin[3], const float bmax[3], const float org[3],
             const float dir[3]);
/* Scene::checkConnectable, include/Scene.h:49-56: closest hit along a -> b, connected
 * iff something is hit and |distance - |b - a|| < 1e-3 (the ESP argument is unused there). */
void orc_check_connectable(const orc_scene *s, const float *a3, const float *b3, int64_t n,
                           uint8_t *connected);

/* per-query visit counters accumulated by orc_intersect since the last reset:
 * [0]=queries [1]=box tests [2]=triangle tests [3]=sphere tests */
void orc_counters(uint64_t out[4], int reset);

/* ---- materials, src/Material.cpp ---- */
void orc_material_eval(const orc_scene *s, int material, const float *wi3,
                       const float *wo3, const float *n3, int64_t count,
                       float *out3);
void orc_material_pdf(const orc_scene *s, int material, const float *wi3,
                      const float *wo3, const float *n3, int64_t count,
                      float *out);
/* u2: the (alph, theta) uniforms the reference would draw, in draw order */
void orc_material_sample(const orc_scene *s, int material, const float *wi3,
                         const float *n3, const float *u2, int64_t count,
                         float *wo3);

/* ---- light sampling: Scene::samplingLight, src/Scene.cpp:16-35 ---- */
/* Draws come from u (in reference draw order); returns number consumed.
 * prim = sampled primitive id (-1 if nothing was sampled). */
int orc_sample_light(const orc_scene *s, const float *u, int n_u, float pos[3],
                     float nrm[3], float *pdf, int32_t *prim);

/* ---- integrator ---- */
/* Render::pathTracing exactly as written (recursive, duplicate re-intersect
 * included), src/Render.cpp:77-130.  Draw order == reference draw order. */
void orc_path_trace(const orc_scene *s, const float org[3], const float dir[3],
                    orc_rng_fn rng, void *user, float rgb[3]);

/* Same, drawing from an explicit array of uniforms (in reference draw order);
 * *used = number of draws consumed (draws past n_u read as 0). */
void orc_path_trace_u(const orc_scene *s, const float org[3], const float dir[3],
                      const float *u, int n_u, float rgb[3], int32_t *used);

/* Philox4x32-10 block, for known-answer tests */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                       uint32_t out[4]);
/* The counter-based stream used by the renderers below: uniform number `dim`
 * of vertex `bounce` of sample `sample` of pixel `pixel`. */
float orc_stream_uniform(uint32_t seed, uint32_t pixel, uint32_t sample,
                         uint32_t bounce, uint32_t dim);

typedef struct {
  int32_t width, height, spp;
  uint32_t seed;
  int32_t chunk_spp;   /* samples summed sequentially per partial sum */
  int32_t max_depth;   /* <=0: unbounded, like the reference */
  float rr_prob;       /* reference: 0.6f */
  int32_t threads;     /* <=0: hardware_concurrency */
  int32_t recursive;   /* 1: reference recursion order; 0: forward throughput */
  int32_t col_begin, col_end; /* render columns [begin,end); end<=0: width */
} orc_render_params;

/* Render::renderPathTracing's pixel/sample loop, src/Render.cpp:632-665, with
 * the counter-based stream in place of the global mt19937.  rgb: W*H*3 floats
 * indexed (i*H+j)*3 like the reference buffer.  stats (may be NULL):
 * [0]=samples [1]=unique rays traced [2]=reference-equivalent intersect calls
 * [3]=path vertices shaded. */
void orc_render(const orc_scene *s, const orc_render_params *p, float *rgb,
                uint64_t stats[4]);
/* primary ray for pixel (i,j), src/Render.cpp:639-655 */
void orc_primary_ray(const orc_scene *s, int W, int H, int i, int j,
                     float org[3], float dir[3]);
/* ---- photon emission: Render::createPhotonMap, src/Render.cpp:405-481 ----
 * The reference fills photonMap[k] sequentially per thread from the shared mt19937.
 * Here emission path number `path` draws bounce b's uniforms from the counter stream
 * (pixel = path, sample = 0, bounce = b) under key (seed, 1), and the map is the first
 * photon_num photons of paths 0, 1, 2, ... in that order -- what one reference thread
 * produces from such draws. */
/* photons path `path` stores (at most `room`, at most 65); returns their number, -1
 * without lights.  Output arrays hold 65 x 3 floats (may be NULL). */
int32_t orc_photon_path(const orc_scene *s, uint32_t seed, uint32_t path, int32_t room, float *pos3,
                        float *wi3, float *power3);
int64_t orc_create_photon_map(const orc_scene *s, uint32_t seed, int64_t photon_num, float *pos3,
                              float *wi3, float *power3, uint32_t *paths_used);
/* the same drawing from an explicit array of uniforms in the reference's draw order
 * (one thread, i.e. photon_num < 16 in the reference); *used = draws consumed */
int32_t orc_create_photon_map_u(const orc_scene *s, const float *u, int n_u, int32_t photon_num,
                                float *pos3, float *wi3, float *power3, int32_t *used);

/* tone map one linear value like Render::saveFile, src/Render.cpp:10-12 */
unsigned char orc_tonemap(float v);

#ifdef __cplusplus
}
#endif
#endif
