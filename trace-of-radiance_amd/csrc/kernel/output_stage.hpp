// kernel/output_stage.hpp -- the small kernels around the integrator: finalize (canvas.nim:47-54), quantize (io/ppm.nim:15-16),
// encode_ipcm (animation output stage), gather_rows (multi-GPU assembly), selftest, spin_until.  Textually included by tor_kernels.hip.
// canvas.nim:47-54
__global__ __launch_bounds__(256) void finalize_kernel(double* pixels, long long n_values, double scale,
                                                        double gamma) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_values) pixels[i] = pow_pos(scale * pixels[i], gamma);
}

// io/ppm.nim:15-16 ; safe_math.nim:10-14
__global__ __launch_bounds__(256) void quantize_kernel(const double* pixels, long long n_values, uint8_t* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_values) {
    double c = pixels[i];
    double cl = (c < 0.0) ? 0.0 : ((c > 0.999) ? 0.999 : c);
    out[i] = (uint8_t)(int)(256 * cl);
  }
}

// ---------------------------------------------------------------------------------------------
// Per-frame output stage of the animation driver (trace_of_radiance_animation.nim:186-196), fused:
//   Canvas -> RGB8        io/rgb.nim:17-31   uint8(256 * clamp(c, 0, 0.999)), top scanline first.
//                         (rgb.nim:29-31 indexes canvas[nrows - i, j], one row past the end for
//                         i = 0; the intended flip canvas[nrows - 1 - i, j] is implemented here.)
//   RGB8 -> Y'CbCr 4:2:0  io/color_conversions.nim:180-252, BT.601 fixed point:
//                         kr,kg,kb = 77,150,29 (>>8); y_scale = 110 (>>7), y_min = 16; fb = 127, fr = 160 (>>8)
//   planes -> I_PCM slice io/h264.nim:189-259: slice header, per macroblock [0x0d 0x00 except the
//                         first] + 256 Y + 64 Cb + 64 Cr raw bytes, stop byte 0x80.
// One workgroup per 16x16 macroblock, one thread per pixel.  Integer and byte work, HBM-bound:
// 24 B read and 1.5 B written per pixel.
__global__ __launch_bounds__(256) void encode_ipcm_kernel(const double* pixels, int nrows, int ncols, uint8_t* out,
                                                          uint8_t* plane_y, uint8_t* plane_cb, uint8_t* plane_cr) {
  __shared__ short s_u[256], s_v[256];
  const int mb_cols = (ncols + 15) >> 4;
  const int mb = blockIdx.x;
  const int mi = mb / mb_cols, mj = mb - mi * mb_cols;
  const int x = threadIdx.x >> 4, y = threadIdx.x & 15;      // row, column inside the macroblock
  const int vr = mi * 16 + x, vc = mj * 16 + y;              // video row (0 = top), column
  // a size that is not a multiple of 16: the last macroblock row / column is padded by edge replication (the SPS crops it)
  const int sr = vr < nrows ? vr : nrows - 1, sc = vc < ncols ? vc : ncols - 1;
  const bool inside = vr < nrows && vc < ncols;
  const double* px = pixels + ((size_t)(nrows - 1 - sr) * ncols + sc) * 3;
  int rgb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double v = px[c];
    const double cl = (v < 0.0) ? 0.0 : ((v > 0.999) ? 0.999 : v);  // safe_math.nim:10-14
    rgb[c] = (int)(uint8_t)(int)(256 * cl);
  }
  const int tY = (77 * rgb[0] + 150 * rgb[1] + 29 * rgb[2]) >> 8;     // color_conversions.nim:218-220
  const uint8_t Y = (uint8_t)(((tY * 110) >> 7) + 16);                // :223
  s_u[threadIdx.x] = (short)(rgb[2] - tY);                            // :221
  s_v[threadIdx.x] = (short)(rgb[0] - tY);                            // :222
  const size_t data = 9 + (size_t)mb * 386;                           // first payload byte of this macroblock
  out[data + threadIdx.x] = Y;
  if (plane_y && inside) plane_y[(size_t)vr * ncols + vc] = Y;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cx = threadIdx.x >> 3, cy = threadIdx.x & 7;
    const int b0 = (2 * cx) * 16 + 2 * cy;
    const int tU = s_u[b0] + s_u[b0 + 1] + s_u[b0 + 16] + s_u[b0 + 17];
    const int tV = s_v[b0] + s_v[b0 + 1] + s_v[b0 + 16] + s_v[b0 + 17];
    const uint8_t U = (uint8_t)((((tU >> 2) * 127) >> 8) + 128);      // :249
    const uint8_t V = (uint8_t)((((tV >> 2) * 160) >> 8) + 128);      // :250
    out[data + 256 + threadIdx.x] = U;
    out[data + 320 + threadIdx.x] = V;
    const size_t cpos = (size_t)(mi * 8 + cx) * (ncols >> 1) + (mj * 8 + cy);
    const bool cinside = mi * 8 + cx < (nrows >> 1) && mj * 8 + cy < (ncols >> 1);
    if (plane_cb && cinside) plane_cb[cpos] = U;
    if (plane_cr && cinside) plane_cr[cpos] = V;
  }
  if (threadIdx.x == 0) {
    if (mb == 0) {  // h264.nim:38: constant slice header (start code, IDR slice NAL, I_PCM first macroblock)
      const uint8_t hdr[9] = {0x00, 0x00, 0x00, 0x01, 0x05, 0x88, 0x84, 0x21, 0xa0};
      for (int k = 0; k < 9; ++k) out[k] = hdr[k];
    } else {        // h264.nim:39,191-192: mb_type I_PCM for every further macroblock
      out[data - 2] = 0x0d;
      out[data - 1] = 0x00;
    }
    if (mb == (int)gridDim.x - 1) out[data + 384] = 0x80;  // h264.nim:40,259: slice stop bit
  }
}

// Multi-GPU assembly: the shards arrive rank-major and compact; put every row at its image position.
// One thread per float64 value; HBM-bound copy (48 B per pixel).
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* gathered, double* frame, int nrows, int ncols,
                                                          int row_tile, int shard_count, long long shard_stride) {
  const long long row_values = (long long)ncols * 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nrows * row_values) return;
  const int row = (int)(i / row_values);
  const long long within_row = i - (long long)row * row_values;
  const int tile = row / row_tile;
  const int shard = tile % shard_count;
  const int local_row = (tile / shard_count) * row_tile + (row - tile * row_tile);
  frame[i] = gathered[(long long)shard * shard_stride + (long long)local_row * row_values + within_row];
}

// keeps its stream busy until the host sets *flag (or max_ticks pass): the stand-in for a collective that never completes
__global__ void spin_until_kernel(volatile unsigned* flag, unsigned long long max_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (*flag == 0u && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(127);
}

__global__ void selftest_kernel(int op, const double* x, const double* y, double* out0, double* out1,
                                long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  selftest_math_one(op, x[i], y ? y[i] : 0.0, out0[i], out1 ? out1[i] : out0[i]);
}

