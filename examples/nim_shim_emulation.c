/* nim_shim_emulation.c -- plays the C code Nim's backend generates for the shim of INTEGRATION.md section 2.
 *
 * There is no Nim toolchain in this image, so the {.importc.} shim cannot be compiled here.  What CAN be
 * checked is the thing that shim relies on: a translation unit that has never seen include/tor_render.h,
 * declares its OWN structs (the layouts Nim derives for the reference's value types on x86-64, under Nim's
 * own mangled names) and its OWN prototypes (what `proc tor_render(...) {.importc, cdecl.}` without a
 * `header` pragma makes Nim emit), links against -ltor_mi355x and renders the reference's main()
 * (trace_of_radiance.nim:26-71) to an ASCII PPM on stdout.  If a struct here disagreed with the library's
 * view by a single byte, the image would not be the reference's image.
 *
 *   gcc -O2 examples/nim_shim_emulation.c -L trace-of-radiance_amd/lib -ltor_mi355x -lm -o nim_shim_emulation
 *   ./nim_shim_emulation [value|ptr] > image.ppm       (value: tor_render, HittableList by value;
 *                                                       ptr: tor_render_ptr, everything by pointer)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double NF;     /* Nim float64 */
typedef int64_t NI;    /* Nim int on x86-64 */
typedef int32_t NI32;
typedef float NF32;
typedef uint8_t NU8;

/* primitives/vec3s.nim:12-14 */
typedef struct tyObject_Vec3__9bK { NF x, y, z; } tyObject_Vec3;

/* physics/core.nim:16-28 -- object variant: discriminator first, then the union of the branches */
typedef struct tyObject_Lambertian__aa { tyObject_Vec3 albedo; } tyObject_Lambertian;
typedef struct tyObject_Metal__bb { tyObject_Vec3 albedo; NF fuzz; } tyObject_Metal;
typedef struct tyObject_Dielectric__cc { NF refraction_index; } tyObject_Dielectric;
typedef struct tyObject_Material__dd {
  NU8 kind;
  union {
    struct { tyObject_Lambertian fLambertian; } _kind_1;
    struct { tyObject_Metal fMetal; } _kind_2;
    struct { tyObject_Dielectric fDielectric; } _kind_3;
  };
} tyObject_Material;

/* physics/hittables/spheres.nim:15-18, moving_spheres.nim:15-20 */
typedef struct tyObject_Sphere__ee { tyObject_Vec3 center; NF radius; tyObject_Material material; } tyObject_Sphere;
typedef struct tyObject_MovingSphere__ff {
  tyObject_Vec3 center0, center1;
  NF time0, time1;
  NF radius;
  tyObject_Material material;
} tyObject_MovingSphere;

/* physics/hittables/hittables_variants.nim:50-57 */
typedef struct tyObject_HittableVariant__gg {
  NU8 kind;
  union {
    struct { tyObject_Sphere fSphere; } _kind_1;
    struct { tyObject_MovingSphere fMovingSphere; } _kind_2;
  };
} tyObject_HittableVariant;

/* physics/hittables/hittables_lists.nim:20-24 */
typedef struct tyObject_HittableList__hh { NI len; tyObject_HittableVariant* objects; } tyObject_HittableList;

/* physics/cameras.nim:15-22 */
typedef struct tyObject_Camera__ii {
  tyObject_Vec3 origin, lower_left_corner, horizontal, vertical, u, v, w;
  NF lens_radius, shutterOpen, shutterClose;
} tyObject_Camera;

/* primitives/canvas.nim:20-28 */
typedef struct tyObject_Canvas__jj {
  tyObject_Vec3* pixels;
  NI32 nrows, ncols;
  NI32 samples_per_pixel;
  NF32 gamma_correction;
} tyObject_Canvas;

/* `static: doAssert sizeof(X) == N` of the shim */
_Static_assert(sizeof(tyObject_Material) == 40, "Material");
_Static_assert(sizeof(tyObject_Sphere) == 72, "Sphere");
_Static_assert(sizeof(tyObject_MovingSphere) == 112, "MovingSphere");
_Static_assert(sizeof(tyObject_HittableVariant) == 120, "HittableVariant");
_Static_assert(sizeof(tyObject_HittableList) == 16, "HittableList");
_Static_assert(sizeof(tyObject_Camera) == 192, "Camera");
_Static_assert(sizeof(tyObject_Canvas) == 24, "Canvas");

/* what Nim emits for the importc procs of the shim (no header pragma -> its own prototypes) */
extern int tor_render(tyObject_Canvas* canvas, tyObject_Camera* cam, tyObject_HittableList world, long long max_depth);
extern int tor_render_ptr(tyObject_Canvas* canvas, tyObject_Camera* cam, tyObject_HittableList* world, long long max_depth);
extern const char* tor_last_error(void);
/* host-side helpers standing in for the reference's own random_scene / camera / exportToPPM (they run in Nim) */
extern long long tor_random_scene(unsigned long long seed, tyObject_HittableVariant* out, long long cap);
extern int tor_camera_init(tyObject_Camera* out, const tyObject_Vec3* look_from, const tyObject_Vec3* look_at,
                           const tyObject_Vec3* view_up, double vfov, double aspect, double aperture, double focus,
                           double shutter_open, double shutter_close);
extern int tor_canvas_to_rgb8(const tyObject_Canvas* canvas, unsigned char* out);

int main(int argc, char** argv) {
  const int by_ptr = argc > 1 && strcmp(argv[1], "ptr") == 0;
  const int width = 384, height = (int)(384 / (16.0 / 9.0)), spp = 100, max_depth = 50; /* trace_of_radiance.nim:27-32 */
  tyObject_HittableVariant* objects = (tyObject_HittableVariant*)calloc(2048, sizeof *objects);
  const long long n = tor_random_scene(0xFACADEull, objects, 2048);                      /* :34-36 */
  if (n <= 0) { fprintf(stderr, "random_scene failed\n"); return 1; }
  /* touch the objects through THIS unit's view of the layout: object 0 is the ground sphere (scenes.nim:15) */
  if (objects[0].kind != 0 || objects[0]._kind_1.fSphere.radius != 1000.0 || objects[0]._kind_1.fSphere.material.kind != 0 ||
      objects[1].kind != 1 || objects[1]._kind_2.fMovingSphere.time1 != 1.0) {
    fprintf(stderr, "layout mismatch: the library's objects do not read back through the Nim-side structs\n");
    return 2;
  }
  const tyObject_Vec3 from = {13, 2, 3}, at = {0, 0, 0}, vup = {0, 1, 0};                /* :38-43 */
  tyObject_Camera cam;
  tor_camera_init(&cam, &from, &at, &vup, 20.0, 16.0 / 9.0, 0.1, 10.0, 0.0, 1.0);        /* :45-51 */
  tyObject_Canvas canvas;                                                                /* :53-57, canvas.nim:30-41 */
  canvas.pixels = (tyObject_Vec3*)malloc((size_t)width * height * sizeof(tyObject_Vec3));
  canvas.nrows = height; canvas.ncols = width; canvas.samples_per_pixel = spp; canvas.gamma_correction = 2.2f;
  tyObject_HittableList world = {n, objects};                                            /* hittables_lists.nim:41-46 */
  const int ok = by_ptr ? tor_render_ptr(&canvas, &cam, &world, max_depth)               /* render.nim:49 */
                        : tor_render(&canvas, &cam, world, max_depth);
  if (ok != 0) { fprintf(stderr, "render failed: %s\n", tor_last_error()); return 1; }   /* doAssert ok == 0 */
  unsigned char* rgb = (unsigned char*)malloc((size_t)width * height * 3);               /* io/ppm.nim:14-27 */
  tor_canvas_to_rgb8(&canvas, rgb);
  printf("P3\n%d %d\n255\n", width, height);
  for (size_t i = 0; i < (size_t)width * height * 3; i += 3) printf("%d %d %d\n", rgb[i], rgb[i + 1], rgb[i + 2]);
  return 0;
}
