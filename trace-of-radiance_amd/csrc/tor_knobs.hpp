// tor_knobs.hpp -- THE table of environment knobs of libtor_mi355x.so.  Every getenv of the library goes through tor::knob(),
// and every name passed to it must be in this table: tests/test_knobs.py greps the sources for both rules, KNOBS.md is generated
// from the table (tools/gen_knob_doc.py, checked by the same test), tor_knob_count / tor_knob_info expose it over the C ABI.
// None of the knobs changes a pixel except TOR_DEFAULT_SEEDING (a different, equally valid sample set) and TOR_FAULT_INJECT /
// negative TOR_SRV_STALL_S (test settings that exercise failure paths; the canvas stays right).
//   when : "call"    read at every tor_render / tor_render_opt / tor_render_device call
//          "context" read once, at tor_context_create (the drop-in's cached contexts: first use of a device in the process)
//          "upload"  read when a scene's culling layout is built
#pragma once

#include <cstdlib>
#include <cstring>

namespace tor {

struct Knob {
  const char* name;
  const char* dflt;
  const char* range;
  const char* when;
  const char* what;
};

constexpr Knob kKnobs[] = {
    // ---- the drop-in's defaults: a host that keeps render()'s signature has no TorOptions ----
    {"TOR_DEFAULT_ACCEL", "3", "0..3", "call", "TOR_ACCEL_* bits tor_render() runs with (3 = block culling + float32 pre-filter, both exact); 0 = the reference's float64 brute force"},
    {"TOR_DEFAULT_SEEDING", "pixel", "pixel | sample", "call", "random streams of tor_render(): the reference's one stream per pixel (render.nim:59-67) or the counter-based per-sample streams (TOR_SEED_SAMPLE: the mode that shards a frame over 8 GPUs)"},
    {"TOR_DEVICES", "(current device)", "all | comma list of HIP ordinals", "call", "device list of tor_render(): row shards on several GPUs, framebuffer gather inside the library"},
    {"TOR_GATHER", "auto", "auto | rccl | peer | host", "call", "how the row shards of a device list reach the canvas (TorOptions.gather)"},
    // ---- multi-GPU gather: watchdog and test hooks ----
    {"TOR_RCCL_TIMEOUT_MS", "10000 + 1 per MB", "> 0", "call", "deadline of one RCCL framebuffer transfer; past it the communicators are aborted and TOR_GATHER=auto carries on with peer copies"},
    {"TOR_RCCL_INIT_TIMEOUT_MS", "120000", "> 0", "call", "deadline of communicator creation + self-check (runs in a helper thread that is abandoned when it does not return)"},
    {"TOR_FAULT_INJECT", "(none)", "comma list of rccl_init | rccl_xfer | rccl_hang | peer", "call", "TEST: the named gather leg fails (or, rccl_hang, never completes) at that point"},
    // ---- launch shape ----
    {"TOR_BLOCKS_PER_CU", "3", "1..8", "context", "workgroups per CU of integrate_kernel (all modes)"},
    {"TOR_WAVES_PER_SIMD", "(from the launch shape)", "2 | 3", "context", "force the register-budget variant of integrate_kernel (256 / 168 VGPRs)"},
    {"TOR_ACCEL_ORDER", "sah", "sah | morton", "upload", "order of the objects behind TOR_ACCEL_BLOCKS' boxes: top-down surface-area build (round 6) or the Morton curve of rounds 1-5 (A/B; same canvas)"},
    {"TOR_STAGE_LDS", "(no cap)", "bytes, 0 = off", "call", "cap of the LDS staging of block records / boxes (TOR_ACCEL_BLOCKS)"},
    {"TOR_SCREEN", "1", "0 | 1", "context", "0: strict brute-force launches evaluate the reference's unfused discriminant for every object instead of the conservative FMA screen (same canvas)"},
    {"TOR_PLANE", "1", "0 | 1 | 2", "context", "stage one of the FMA screen (the 4-instruction plane screen in front of the segment's wave-uniform test, per-lane stage two on what it keeps): 0 off, 1 on the segments where it pays (default), 2 on every segment (same candidates, same canvas)"},
    // ---- TOR_SEED_PIXEL: kernel choice, cost probe, tile schedule ----
    {"TOR_COOP_MAX_PIXELS", "114688", ">= 0", "context", "frames up to this many pixels (per device) run one WAVE per pixel when neither hand-off nor split mode applies; 0 = never"},
    {"TOR_LPT_MIN_SPP", "32", ">= 0", "context", "cost probe + chain-length-ordered tile schedule from this many spp on; 0 = never"},
    {"TOR_HOT_FRAC", "0.4", ">= 0", "context", "a pixel chain is HOT (arbiter priority 3) from this share of an average wave's iterations; 0 = off"},
    {"TOR_PRIO_SHIFT", "16", "0..31", "context", "arbiter-priority rotation period, log2 shader-clock ticks; 0 = off"},
    // ---- TOR_SEED_PIXEL chain hand-off (DESIGN 4.7 (HISTORY 4.10)) ----
    {"TOR_MIGRATE", "1", "0 | 1", "context", "0: no chain hand-off (split mode / wave-per-pixel kernel instead)"},
    {"TOR_SRV_FRAC", "0.07", "0..1", "context", "share of the workgroups that start as servers when the frame can hold a chain above the threshold's floor"},
    {"TOR_SRV_MIN_FRAC", "0.005", "0..1", "context", "... otherwise"},
    {"TOR_SRV_PATIENCE_US", "8000", ">= 0", "context", "a dedicated server without a chain for this long becomes a lane wave; 0 = never"},
    {"TOR_SRV_STALL_S", "60", "seconds; 0 = never; < 0 = at once (TEST)", "call", "a WAITING server that sees no progress of the frame for this long flags the frame incomplete and leaves (launch not fully resident); the blocking entry points render it again without the hand-off"},
    {"TOR_PUSH_THETA", "3.0", "> 0", "context", "first push threshold, x the bounce iterations of an average lane"},
    {"TOR_FLOOR_THETA", "1.33", "> 0", "context", "floor of the adaptive push threshold, same unit"},
    {"TOR_CHAIN_THETA", "3.5", "> 0", "context", "... and never below this x the frame's mean chain"},
    {"TOR_TAIL_LANES", "8", "-2..64", "context", "an exhausted wave hands its chains over from this many live lanes down; -1 nobody pushes in the tail; -2 nobody serves (debug)"},
    {"TOR_TAIL_REST", "256", ">= 0", "context", "... with more live lanes: only chains with at least this many bounce iterations to go"},
    {"TOR_MIG_FLAGS", "0x806", "bit mask", "context", "bit 0 acquire polling, bit 1 adaptive threshold, bit 2 tail chains served at priority 1, bits 8-15 longest back-off (naps), bits 16-31 cap of waiting servers"},
};

constexpr int kKnobCount = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));

// getenv for a name of the table.  A name that is not in the table is a programming error: it reads as unset (and
// tests/test_knobs.py fails on the source).
inline const char* knob(const char* name) {
  for (int i = 0; i < kKnobCount; ++i)
    if (std::strcmp(kKnobs[i].name, name) == 0) return std::getenv(name);
  return nullptr;
}

}  // namespace tor
