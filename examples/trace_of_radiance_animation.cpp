// trace_of_radiance_animation.cpp -- the reference's video driver (trace_of_radiance_animation.nim:
// main_animation_mp4, :101-214, "fast test" constants :107-119) on top of the C ABI of
// libtor_mi355x.so: bouncing-spheres animation, one render per frame, RGB -> Y'CbCr 4:2:0 -> I_PCM
// H.264 on the device, an Annex-B `animation.264` on disk, then muxed into `animation.mp4`
// (MP4Muxer, :203-210; 30 frames per second as io/mp4.nim:141).
//
//   g++ -O2 -I include examples/trace_of_radiance_animation.cpp -L trace-of-radiance_amd/lib \
//       -ltor_mi355x -Wl,-rpath,'$ORIGIN/../trace-of-radiance_amd/lib' -o examples/trace_of_radiance_animation
//   ./examples/trace_of_radiance_animation [out.264] [width] [spp] [t_max]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "tor_render.h"

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "animation.264";
  const double aspect_ratio = 16.0 / 9.0;                                // :108
  const int image_width = argc > 2 ? std::atoi(argv[2]) : 256;           // :109 (sizes that are not multiples of 16 are padded and cropped: tor_h264_frame_bytes)
  const int image_height = (int)(image_width / aspect_ratio);            // :110
  const int samples_per_pixel = argc > 3 ? std::atoi(argv[3]) : 10;      // :111
  const float gamma_correction = 2.2f;                                   // :112
  const int max_depth = 50;                                              // :113
  const float dt = 0.005f, t_min = 0.0f;                                 // :116-117
  const float t_max = argc > 4 ? (float)std::atof(argv[4]) : 1.0f;       // :118
  const int skip = 6;                                                    // :119

  TorAnimation* anim = nullptr;
  if (tor_animation_create(0xFACADE, image_height, image_width, dt, t_min, t_max, &anim) != TOR_OK) return 1;  // :141-146
  TorContext* ctx = nullptr;
  if (tor_context_create(-1, &ctx) != TOR_OK) { std::fprintf(stderr, "%s\n", tor_last_error()); return 1; }

  std::FILE* out = std::fopen(path, "wb");                               // :158-160
  if (!out) return 1;
  uint8_t header[64];
  const int hn = tor_h264_stream_header(image_width, image_height, header, sizeof header);  // H264Encoder.init
  const int64_t fn = tor_h264_frame_bytes(image_width, image_height);
  if (hn < 0 || fn < 0) { std::fprintf(stderr, "width and height must be even (4:2:0)\n"); return 1; }
  std::fwrite(header, 1, (size_t)hn, out);

  TorOptions opt{};
  opt.struct_size = sizeof(TorOptions);
  opt.seeding = TOR_SEED_PIXEL;        // the reference's streams
  opt.arith = TOR_ARITH_STRICT;
  opt.device = -1;
  opt.shard_count = 1;
  opt.row_tile = 1;
  opt.accel = TOR_ACCEL_BLOCKS | TOR_ACCEL_F32;  // ~1600 spheres per frame: both exact accelerations (what tor_render() defaults to), identical pixels

  std::vector<TorHittableVariant> objects((size_t)tor_animation_object_count(anim));
  std::vector<uint8_t> slice((size_t)fn);
  TorCamera cam;
  int64_t n_objects = 0;
  float t = 0.f;
  int scene_id = 0;
  const auto start = std::chrono::steady_clock::now();
  while (tor_animation_next(anim, skip, &cam, objects.data(), (int64_t)objects.size(), &n_objects, &t) == 1) {  // :173
    if (tor_scene_upload(ctx, TorHittableList{n_objects, objects.data()}) != TOR_OK ||
        tor_render_frame_h264(ctx, &cam, image_height, image_width, samples_per_pixel, gamma_correction, max_depth, &opt,
                              slice.data(), (int64_t)slice.size()) != TOR_OK) {                               // :181-196
      std::fprintf(stderr, "frame %d failed: %s\n", scene_id, tor_last_error());
      return 1;
    }
    std::fwrite(slice.data(), 1, slice.size(), out);
    ++scene_id;
  }
  const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
  std::fclose(out);
  tor_context_destroy(ctx);
  tor_animation_destroy(anim);
  std::fprintf(stderr, "%d frames (%dx%d, %d spp) in %.3f s -> %s\n", scene_id, image_width, image_height, samples_per_pixel,
               elapsed, path);
  // Muxing into MP4 (:203-210): <name>.264 -> <name>.mp4
  std::string mp4 = path;
  const size_t dot = mp4.rfind('.');
  mp4 = (dot == std::string::npos ? mp4 : mp4.substr(0, dot)) + ".mp4";
  const int samples = tor_mp4_mux_file(path, mp4.c_str(), image_width, image_height, 30);
  if (samples != scene_id) {
    std::fprintf(stderr, "muxing failed: %s\n", tor_last_error());
    return 1;
  }
  std::fprintf(stderr, "Finished! Rendering available at \"%s\"\n", mp4.c_str());
  return 0;
}
