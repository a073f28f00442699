// Command-line equivalent of the reference's driver (src/main.cpp:5-101): builds
// the same Cornell-box scene through the same API calls and renders it with
// path tracing on the GPU.
//   rtr_main [width height spp [output-basename [seed [devices]]]]
// writes <output-basename>.ppm in the reference's format (src/Render.cpp:3-21).
// devices: comma-separated GPU indices, e.g. 0,1,2,3,4,5,6,7 (a GPU may repeat); the image
// is then rendered as one partition per entry and gathered on the first (RenderOptions::devices).
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "rtr_host.hpp"

int main(int argc, char **argv) {
  unsigned W = argc > 1 ? (unsigned)atoi(argv[1]) : 512;
  unsigned H = argc > 2 ? (unsigned)atoi(argv[2]) : 512;
  unsigned spp = argc > 3 ? (unsigned)atoi(argv[3]) : 1;
  std::string out = argc > 4 ? argv[4] : "test0";
  unsigned seed = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
  std::vector<int> devices;
  if (argc > 6) {
    std::stringstream list(argv[6]);
    for (std::string item; std::getline(list, item, ',');) devices.push_back(atoi(item.c_str()));
  }

  Material *whiteLightMaterial = new Material(
      MaterialType::DIFFUSE, 8.0f * Vector3f(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) +
                                 15.6f * Vector3f(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                                 18.4f * Vector3f(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));
  auto diffuse = [](float r, float g, float b) {
    return new Material(MaterialType::DIFFUSE, Vector3f(0), Vector3f(r, g, b));
  };
  Material *white = diffuse(0.6, 0.6, 0.6), *red = diffuse(0.6, 0.01, 0.01),
           *green = diffuse(0.01, 0.6, 0.01), *blue = diffuse(0.01, 0.01, 0.6),
           *yellow = diffuse(0.6, 0.6, 0.01), *pink = diffuse(0.01, 0.6, 0.6),
           *cyan = diffuse(0.6, 0.01, 0.6);
  Material *specular = new Material(MaterialType::SPECULAR, Vector3f(0), Vector3f(1));
  Material *microfacet = new Material(MaterialType::MICROFACET, Vector3f(0),
                                      Vector3f(0.6f, 0.01f, 0.01f), Vector3f(0.6f, 0.01f, 0.01f), 0, 0.5f);

  std::vector<Vector3f> cube = {Vector3f(0, 0, -512), Vector3f(512, 0, -512), Vector3f(0, 512, -512),
                                Vector3f(512, 512, -512), Vector3f(0, 0, 0), Vector3f(512, 0, 0),
                                Vector3f(0, 512, 0), Vector3f(512, 512, 0)};
  struct Wall { Material *m; std::vector<unsigned int> idx; };
  std::vector<Wall> walls = {{red, {0, 1, 2, 3, 2, 1}},    {white, {0, 4, 1, 5, 1, 4}},
                             {yellow, {2, 3, 6, 7, 6, 3}}, {pink, {0, 2, 4, 6, 4, 2}},
                             {cyan, {1, 5, 3, 7, 3, 5}}};

  Camera *camera = new Camera(Vector3f(256, 256, 512), Vector3f(0, 0, -1), Vector3f(0, 1, 0), 45);
  Scene scene(camera);
  scene.addObject(new Sphere(whiteLightMaterial, Vector3f(256, 256, -256), 20));
  for (Wall &w : walls) scene.addObject(new TrangleMesh(w.m, &cube, &w.idx));
  scene.addObject(new Sphere(red, Vector3f(256, 512 - 64, -256), 32));
  scene.addObject(new Sphere(blue, Vector3f(64, 256, -256), 32));
  scene.addObject(new Sphere(green, Vector3f(512 - 64, 256, -256), 32));
  scene.addObject(new Sphere(microfacet, Vector3f(256, 64, -256), 32));
  scene.addObject(new Sphere(specular, Vector3f(64, 64, -512 + 64), 64));
  scene.addObject(new Sphere(specular, Vector3f(64, 512 - 64, -512 + 64), 64));
  scene.addObject(new Sphere(specular, Vector3f(512 - 64, 64, -512 + 64), 64));
  scene.addObject(new Sphere(specular, Vector3f(512 - 64, 512 - 64, -512 + 64), 64));
  scene.rebuildBVH();

  Render renderer;
  renderer.options.seed = seed;
  renderer.options.devices = devices;
  try {
    renderer.render(scene, W, H, spp, out, RenderType::PathTracing);
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << "\n";
    return 1;
  }
  const rtr_stats &st = renderer.getStats();
  std::cout << "samples " << st.samples << "  rays " << st.rays << "  " << st.ms_total << " ms  "
            << (st.samples / st.ms_total / 1e3) << " Msamples/s on " << st.n_devices << " device(s)\n";
  return 0;
}
