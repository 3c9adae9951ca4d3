// trace_of_radiance_main.cpp -- the reference's default executable (trace_of_radiance.nim:26-71)
// on top of the C ABI of libtor_mi355x.so: same constants, same scene seed, same camera,
// wall-clock bracket around render() only, ASCII PPM on stdout, "Time spent" on stderr.
//
//   g++ -O2 -I include examples/trace_of_radiance_main.cpp -L trace-of-radiance_amd/lib \
//       -ltor_mi355x -Wl,-rpath,'$ORIGIN/../trace-of-radiance_amd/lib' -o examples/trace_of_radiance
//   ./examples/trace_of_radiance [width] [spp] [sample|pixel] > image.ppm
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tor_render.h"

int main(int argc, char** argv) {
  const double aspect_ratio = 16.0 / 9.0;                                   // :27
  const int image_width = argc > 1 ? std::atoi(argv[1]) : 384;              // :28
  const int image_height = (int)(image_width / aspect_ratio);               // :29
  const int samples_per_pixel = argc > 2 ? std::atoi(argv[2]) : 100;        // :30
  const float gamma_correction = 2.2f;                                      // :31
  const int max_depth = 50;                                                 // :32
  const bool per_sample = argc > 3 && std::strcmp(argv[3], "sample") == 0;

  std::vector<TorHittableVariant> world(2048);
  const int64_t n = tor_random_scene(0xFACADE, world.data(), (int64_t)world.size());  // :34-36
  if (n < 0) { std::fprintf(stderr, "random_scene failed\n"); return 1; }

  const TorVec3 look_from{13, 2, 3}, look_at{0, 0, 0}, vup{0, 1, 0};        // :38-43
  TorCamera cam;
  tor_camera_init(&cam, &look_from, &look_at, &vup, 20.0, aspect_ratio, 0.1, 10.0, 0.0, 1.0);  // :45-51

  std::vector<TorVec3> pixels((size_t)image_height * image_width);          // :53-57
  TorCanvas canvas{pixels.data(), image_height, image_width, samples_per_pixel, gamma_correction};

  TorOptions opt{};
  opt.struct_size = sizeof(TorOptions);
  opt.seeding = per_sample ? TOR_SEED_SAMPLE : TOR_SEED_PIXEL;
  opt.arith = TOR_ARITH_STRICT;
  opt.device = -1;
  opt.shard_count = 1;
  opt.row_tile = 1;

  const auto start = std::chrono::steady_clock::now();                      // :60
  const int rc = tor_render_opt(&canvas, &cam, TorHittableList{n, world.data()}, max_depth, &opt);  // :62
  const auto stop = std::chrono::steady_clock::now();                       // :64
  if (rc != TOR_OK) { std::fprintf(stderr, "render failed: %s\n", tor_last_error()); return 1; }

  std::vector<uint8_t> rgb((size_t)image_height * image_width * 3);         // io/ppm.nim:14-27
  tor_canvas_to_rgb8(&canvas, rgb.data());
  std::printf("P3\n%d %d\n255\n", image_width, image_height);
  for (size_t i = 0; i < rgb.size(); i += 3) std::printf("%d %d %d\n", rgb[i], rgb[i + 1], rgb[i + 2]);
  const double elapsed = std::chrono::duration<double>(stop - start).count();
  std::fprintf(stderr, "\nDone.\nTime spent: %6.3f s  (%.1f Msamples/s incl. upload and read-back)\n", elapsed,
               (double)image_height * image_width * samples_per_pixel / elapsed / 1e6);
  return 0;
}
