// kernel/tile_sort.hpp -- tile_key_kernel / tile_order_kernel / tile_scatter_kernel: the SEED_PIXEL tile schedule (counting sort
// of the tiles by probed chain length, the cuts, the hand-off's schedule words).  Textually included by tor_kernels.hip.
// Tile schedule for SEED_PIXEL: counting sort of the tiles by probed cost, most expensive first
// (longest-processing-time-first: a pixel is a sequential chain of spp samples, so the expensive
// chains must start at t = 0).  One workgroup; the order of equal-cost tiles is irrelevant (the
// schedule never changes a pixel's value).
constexpr int kCostBins = 4096;

// The probe counts closest-hit queries per PIXEL (2 samples).  A tile's sort key is led by its longest pixel chain --
// lanes pull pixels one by one, so the chain, not the tile's sum, is what has to start early (a tile on the rim of a
// glass sphere has a few 27-bounce pixels among sky: by its sum it would start mid-frame and its chains would end the
// frame) -- with the tile's sum as the tie-breaker; the sum itself is kept for the work accounting of the cuts.
__global__ __launch_bounds__(256) void tile_key_kernel(const unsigned* pixel_cost, unsigned n_pixels, int n_tiles, unsigned* key,
                                                        unsigned* work, unsigned key_mode, unsigned probe_spp, unsigned* ghist,
                                                        unsigned long long* gwork) {
  const int tile = (int)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
  if (tile >= n_tiles) return;
  const unsigned pl = (unsigned)tile * kTilePixels + (threadIdx.x & 63);
  const unsigned c = pl < n_pixels ? pixel_cost[pl] : 0u;
  unsigned mx = c, sum = c;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned om = (unsigned)__shfl_xor((int)mx, off), os = (unsigned)__shfl_xor((int)sum, off);
    mx = om > mx ? om : mx;
    sum += os;
  }
  if ((threadIdx.x & 63) == 0) {
    if (key_mode == 0) {  // round-2 key: the tile's longest probed pixel, the sum breaks ties
      const unsigned m = mx < 127u ? mx : 127u;        // 2 samples x max_depth 50 <= 100 (more only with a deeper max_depth)
      const unsigned t = (sum >> 8) < 31u ? (sum >> 8) : 31u;
      key[tile] = (m << 5) | t;
    } else {
      // Two classes.  A pixel whose probe samples were ALL deep (>= 28 queries per sample on average: inside glass) is a
      // long chain for certain -- one deep path among ordinary ones is not, half of all tiles hold one -- and its tile must
      // start first: upper half of the key space, by that pixel's count.  Every other tile is ordered by its SUM: 64 pixels
      // x the probe's samples predict the mean cost of the tile's chains well, and the frame should end on the cheapest
      // chains (the tail of a frame is as long as the chains that are started last).
      const unsigned per2 = 2u * mx / probe_spp, sum2 = 2u * sum / probe_spp;  // normalised to 2 probe samples
      if (per2 >= 56u) key[tile] = 2048u + ((per2 < 127u ? per2 : 127u) << 4) + ((sum2 >> 6) < 15u ? (sum2 >> 6) : 15u);
      else key[tile] = sum2 < 2047u ? sum2 : 2047u;
    }
    work[tile] = sum > 0 ? sum : 1u;
    // histogram of the counting sort (tile_order_kernel scans it, tile_scatter_kernel places the tiles): per key, tiles and probed work
    const unsigned b = key[tile] < (unsigned)kCostBins ? key[tile] : (unsigned)kCostBins - 1;
    atomicAdd(ghist + b, 1u);
    atomicAdd(gwork + b, (unsigned long long)(sum > 0 ? sum : 1u));
  }
}

// Counting sort of the tiles by descending key (longest-processing-time-first: a pixel is a sequential chain of spp
// samples, so the long chains must start at t = 0).  One workgroup; the order of equal-key tiles is irrelevant (the
// schedule never changes a pixel's value).  One cut of the sorted list, by probed work:
//  * split_frac > 0: the first K tiles carry split_frac of the work (the longest chains: they go to coop_pixel_kernel,
//    one wave per pixel); the lane kernel's counter is started at tile K, K is written for the wave kernel to read;
//  * sched[0] = hot_chain x the probed total: the chain length (bounce iterations) from which a pixel is HOT (priority 3).
__global__ __launch_bounds__(1024) void tile_order_kernel(const unsigned* ghist, const unsigned long long* gwork, unsigned* goffs, int n_tiles,
                                                          float split_frac, unsigned long long* split_out,
                                                          unsigned long long* lane_counter, float hot_chain,
                                                          unsigned long long* sched, const MigSchedule mig) {
  // (one workgroup, but only over the 4096 bins: the per-tile passes on either side -- histogram in tile_key_kernel, placement
  // in tile_scatter_kernel -- run on the whole machine; round 2 did all three here in 0.49 ms at 1080p)
  __shared__ unsigned hist[kCostBins];
  __shared__ unsigned offs[kCostBins];
  __shared__ unsigned long long bin_work[kCostBins];
  for (int i = threadIdx.x; i < kCostBins; i += blockDim.x) { hist[i] = ghist[i]; bin_work[i] = gwork[i]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    unsigned long long total = 0;
    for (int b = kCostBins - 1; b >= 0; --b) {  // descending key
      offs[b] = run;
      run += hist[b];
      total += bin_work[b];
    }
    // first k tiles of the order that carry `target` work (tiles of one bin count with the bin's mean)
    auto tiles_for = [&](unsigned long long target) {
      unsigned long long cum = 0;
      unsigned k = 0;
      for (int b = kCostBins - 1; b >= 0 && cum < target; --b) {
        if (hist[b] == 0) continue;
        const unsigned long long mean = (bin_work[b] + hist[b] - 1) / hist[b];
        const unsigned long long want = (target - cum + mean - 1) / mean;
        const unsigned take = want < hist[b] ? (unsigned)want : hist[b];
        k += take;
        cum += mean * take;
      }
      return k;
    };
    unsigned k_split = 0;
    if (split_out != nullptr) {
      const unsigned long long target = (unsigned long long)((double)split_frac * (double)total);
      k_split = tiles_for(target);
      *split_out = k_split;
      *lane_counter = (unsigned long long)k_split * kTilePixels;
    }
    // hot chains: hot_chain x the probed total, scaled by the host to bounce iterations of the frame
    if (sched != nullptr) sched[0] = hot_chain > 0.0f ? (unsigned long long)(hot_chain * (float)total) + 1ull : 0ull;
    if (mig.mig != nullptr) {
      // Chain hand-off (integrate_kernel / serve_chains).  l_avg = bounce iterations an average lane runs in this frame.
      // The longest chains (glass: ~34 queries per sample whatever the frame) are a fixed number of iterations, so the
      // smaller l_avg -- a small frame, a row shard of a multi-GPU job -- the larger their share of the frame time.
      // A chain is handed over once its projected length exceeds the (adaptive) threshold: push_theta x l_avg at first.
      const float l_avg = (float)total * mig.lavg_scale;
      // push threshold: push_theta x l_avg, but never below chain_theta x the frame's MEAN chain -- on a frame with fewer
      // pixels than lanes l_avg says nothing about how long a chain is
      const float mean_chain = (float)total * mig.chain_scale;
      float push = mig.push_theta * l_avg;
      if (push < mig.chain_theta * mean_chain) push = mig.chain_theta * mean_chain;
      if (push < 64.0f) push = 64.0f;
      // the adaptive threshold starts at `push` and moves between it ... and the length from which a chain cannot finish in a lane
      // before the frame does (floor_theta x l_avg; same floor from the mean chain as above)
      float fl = mig.floor_theta * l_avg;
      if (fl < mig.chain_theta * mean_chain) fl = mig.chain_theta * mean_chain;
      if (fl < 64.0f) fl = 64.0f;
      if (fl > push) fl = push;
      // Dedicated servers.  The share of a frame's work that sits in chains above the threshold is a property of the scene
      // (glass: ~1 %), and serving it costs 64 x that share x (server bounce / lane bounce = ~3 us / 16 us) of the machine
      // whatever the frame size: srv_frac of the workgroups start as servers whenever a chain of this frame CAN reach the
      // threshold at all (a sample has at most max_depth queries; the longest chains of a scene with glass run at ~0.7 of
      // that).  Servers that find nothing to do turn into lane waves after `mig_patience` (serve_chains), so a scene without
      // long chains pays ~patience x srv_frac once.
      float frac = mig.srv_frac;
      if ((float)mig.spp * 0.7f * (float)mig.max_depth < fl) frac = mig.srv_min_frac;
      int n_srv = (int)(frac * (float)mig.blocks + 0.999f);
      // waves that never get a tile (frames with fewer tiles than waves) are servers from the start anyway
      const int free_wgs = (mig.blocks * (kThreads / 64) - n_tiles) / (kThreads / 64);
      if (free_wgs > 0) n_srv = n_srv > free_wgs ? n_srv - free_wgs : 0;
      if (n_srv > mig.blocks - 1) n_srv = mig.blocks - 1;
      if (n_srv < 0) n_srv = 0;
      mig.mig[kMigTCounterDry] = ~0ull;
      mig.mig[kMigSrvWgs] = (unsigned long long)n_srv;
      mig.mig[kMigLaneWaves] = (unsigned long long)(mig.blocks - n_srv) * (kThreads / 64);
      mig.mig[kMigPush] = (unsigned long long)push;
      mig.mig[kMigPushFloor] = (unsigned long long)fl;
      mig.mig[kMigPushNow] = (unsigned long long)push;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kCostBins; i += blockDim.x) goffs[i] = offs[i];
}

// placement pass of the counting sort: tile i goes to the next free position of its key's range
__global__ __launch_bounds__(256) void tile_scatter_kernel(const unsigned* key, unsigned* goffs, unsigned* order, int n_tiles) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n_tiles) return;
  const unsigned b = key[i] < (unsigned)kCostBins ? key[i] : (unsigned)kCostBins - 1;
  order[atomicAdd(goffs + b, 1u)] = (unsigned)i;
}

