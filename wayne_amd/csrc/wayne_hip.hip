// libwayne_hip.so -- C ABI (include/wayne_hip.h) over the gfx950 kernels.
// Host side: context, HBM buffers, launch sequence, HIP-event profiling.
#include "../../include/wayne_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "host_plan.h"

using namespace wayne;

namespace {

enum ProfKernel { PK_PREP_WL = 0, PK_PREP_SUB, PK_THROW, PK_COSMIC, PK_RAMP, PK_LIGHTCURVE, PK_NARROW, PK_LANE };
const char* const kProfNames[WAYNE_PROF_KERNELS] = {"k_prep_wl", "k_prep_sub",   "k_throw",  "k_cosmic",   /* (cosmic rays ride in k_prep_sub: slot kept for the ABI) */
                                                    "k_ramp",    "k_lightcurve", "k_narrow", "k_lane"};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool owns = true;        // false: a view into another allocation (the slot's input arena)
  hipError_t reserve(size_t bytes) {
    if (!owns) { p = nullptr; cap = 0; owns = true; }
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    if (bytes == 0) return hipSuccess;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void view(void* at) {
    if (owns && p) (void)hipFree(p);
    p = at;
    cap = 0;
    owns = false;
  }
  void release() {
    if (p && owns) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    owns = true;
  }
  template <class T> T* as() const { return (T*)p; }
};

struct Slot {
  bool uploaded = false;
  bool front_done = false;
  bool acc_dirty = false;
  bool acc_init = false;
  bool fused_last = false;    // the last front half ran fused: positions / routing arrays were not written (debug_fetch)
  PrepArgs last_prep{};       // ... and k_prep_sub's arguments of that run
  int last_chunks = 0;
  bool force_throw = false;   // the last run met a bin beyond a lane's reach: run with k_throw (set by check_status)
  bool ran = false;           // a whole run was enqueued whose status word nobody has looked at yet (settle)
  wayne_exposure_desc d{};  // host copy (pointers are NOT valid after upload)
  int W = 0, K = 0, R = 0;
  bool has_depth = false, has_replay_seed = false, has_lc = false, has_lc_hidden = false;
  DevBuf wl, flux, depth, xref, yref, dur, rseed, sread, read_dt, lc_z, lc_hidden, lc_rp;   // views into in_dev (depth: owned when computed by k_lightcurve)
  DevBuf ratio, sigl, sigh, sens, dlam;
  DevBuf counts, nwide, nsplit, nlane, prefix, xpos, ypos, sub, chunk_total, chunk_box, tr;
  DevBuf acc, out, misc;  // misc: [0] total electrons (u64), [1] status (int)
  DevBuf seg;             // cosmic-ray segments (CosmicArgs::seg): zeroed when allocated, k_ramp clears what it reads
  int kb = 1;             // sub-samples a workgroup of k_lane takes (k_lane, "BATCHES")
  bool thin = false;      // few electrons per (workgroup, sub-sample): k_lane flushes from its first-touch list
  bool use_box = false;   // acc_box is valid: k_ramp loads the accumulators of a read only inside it (and where `seg` says)
  int acc_box[16][4] = {{0}};
  DevBuf in_dev;          // device mirror of the staging arena: the descriptor's arrays arrive in ONE copy
  void* pinned = nullptr;  // pinned host copy of `out` (fetch_async / wait), followed by a copy of `misc`
  size_t pinned_cap = 0;
  struct Misc { unsigned long long electrons; int status; int pad; };
  Misc* pinned_misc = nullptr;   // inside `pinned`: a copy into pageable memory would block the caller
  // sky alias tables of this exposure (k_ramp): device copy, pinned host copy, the lam_max they were built for
  DevBuf sky_tab;
  uint32_t* sky_tab_host = nullptr;
  hipEvent_t sky_tab_ev = nullptr;
  bool sky_tab_pending = false;
  std::vector<uint32_t> sky_tab_keys;
  bool sky_alias_on = false, sky_pieces = false;
  uint32_t sky_mask = 0;
  int sky_L = 1;
  float sky_level[16] = {0};
  unsigned char sky_tab0[16] = {0};
  std::vector<double> read_dt_host;
  double lc_p_lo = 0., lc_p_hi = 0.;   // range of lc_rp
  double max_narrow = 0.;   // host estimate: most narrow electrons expected in a bin of the longest sub-sample
  double max_chunk_electrons = 0.;   // host estimate: electrons of the fullest k_lane chunk in the longest sub-sample
  double est_thrown = 0.;   // host estimate of the electrons k_throw handles in the longest sub-sample
  unsigned char chunk_order[kMaxChunks] = {0};   // chunks of kNarrowThreads bins, most electrons first (ThrowArgs::chunk_order)
  unsigned char lane_order[kMaxChunks] = {0};    // chunks of kLaneThreads bins, most electrons first
  // pinned staging arena of the descriptor's arrays: uploads are enqueued from here, so
  // wayne_exposure_upload returns without waiting for the slot's stream to drain
  char* stage = nullptr;
  size_t stage_cap = 0, stage_used = 0;
  hipEvent_t stage_ev = nullptr;
  bool stage_pending = false;
  void release() {
    for (DevBuf* b : {&wl, &flux, &depth, &xref, &yref, &dur, &rseed, &sread, &read_dt, &lc_z, &lc_hidden, &lc_rp, &ratio, &sigl,
                      &sigh, &sens, &dlam, &counts, &nwide, &nsplit, &nlane, &prefix, &xpos, &ypos, &sub, &chunk_total, &chunk_box, &tr, &acc, &out,
                      &misc, &seg, &sky_tab, &in_dev})
      b->release();
    if (sky_tab_host) (void)hipHostFree(sky_tab_host);
    sky_tab_host = nullptr;
    if (sky_tab_ev) (void)hipEventDestroy(sky_tab_ev);
    sky_tab_ev = nullptr;
    sky_tab_pending = false;
    sky_tab_keys.clear();
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    pinned_misc = nullptr;
    pinned_cap = 0;
    if (stage) (void)hipHostFree(stage);
    stage = nullptr;
    stage_cap = stage_used = 0;
    if (stage_ev) (void)hipEventDestroy(stage_ev);
    stage_ev = nullptr;
    stage_pending = false;
  }
};

struct ProfRec {
  int kernel;
  hipEvent_t a, b;
};

// HBM slots per context: every exposure of a batch keeps its inputs, scratch,
// accumulators and reads resident (~215 MB each at 1024^2 x 16 reads; buffers
// are allocated on first upload, so unused slots cost nothing).
constexpr int kSlots = 256;

}  // namespace

constexpr int kStreams = 2;   // exposures in even / odd slots run on different HIP streams

// Tuning and test knobs of a context.  Each is read ONCE from the environment (WAYNE_<NAME>) when the context is
// created; afterwards only wayne_ctx_set_knob changes it -- no entry point looks at the environment of a live context,
// so a thread that edits the environment cannot change what another thread's context launches (and getenv never races
// a setenv).  -1 = not set: the library's own choice.
struct Knobs {
  long long tile_ints = -1;      // LDS ints of a k_throw tile (thrower_lds_ints)
  long long batch = -1;          // sub-samples a k_lane workgroup takes (plan::lane_batches)
  long long thin = -1;           // k_lane's first-touch flush list on (1) / off (0)
  long long no_acc_box = -1;     // 1: k_ramp loads every accumulator (no boxes)
  long long lane_reach = -1;     // electrons a lane takes in an exposure launched without k_throw (lowers kLaneReach: exercises the re-run)
  long long throw_wgs = -1;      // k_throw workgroups per launch
  long long keep_narrow = -1;    // 1: k_narrow is launched even when no bin is expected to qualify
  long long no_fuse = -1;        // 1: never k_lane_fused
  long long fork_narrow = -1;    // 1: k_narrow on a side stream beside k_lane
  long long streams = -1;        // 1: every slot on one stream
  long long upload_timing = -1;  // 1: wayne_ctx_destroy prints the host time of wayne_exposure_upload by part
  long long ramp_reads = -1;     // TIMING BUILDS (-DWAYNE_TIMING_KNOBS) only: k_ramp works through the first n reads
};
struct KnobName { const char* name; const char* env; long long Knobs::*field; };
const KnobName kKnobNames[] = {
    {"tile_ints", "WAYNE_TILE_INTS", &Knobs::tile_ints},       {"batch", "WAYNE_BATCH", &Knobs::batch},
    {"thin", "WAYNE_THIN", &Knobs::thin},                      {"no_acc_box", "WAYNE_NO_ACC_BOX", &Knobs::no_acc_box},
    {"lane_reach", "WAYNE_LANE_REACH", &Knobs::lane_reach},    {"throw_wgs", "WAYNE_THROW_WGS", &Knobs::throw_wgs},
    {"keep_narrow", "WAYNE_KEEP_NARROW", &Knobs::keep_narrow}, {"no_fuse", "WAYNE_NO_FUSE", &Knobs::no_fuse},
    {"fork_narrow", "WAYNE_FORK_NARROW", &Knobs::fork_narrow}, {"streams", "WAYNE_STREAMS", &Knobs::streams},
    {"upload_timing", "WAYNE_UPLOAD_TIMING", &Knobs::upload_timing}, {"ramp_reads", "WAYNE_RAMP_READS", &Knobs::ramp_reads},
};
constexpr size_t kMiscBytes = 64;   // status block of a slot: [0] electrons (u64), [8] status (int); k_prep_wl clears all of it

constexpr size_t kCounterBytes = (size_t)kCounterStripes * kCounterStride * sizeof(unsigned long long);

struct wayne_ctx {
  int device = 0;
  hipStream_t stream = nullptr;          // stream of the call in progress (one of streams[])
  hipStream_t streams[kStreams] = {nullptr, nullptr};
  // k_narrow of an exposure runs beside its k_throw (both only add into the accumulators): a side
  // stream per main stream, forked after the prep kernels and joined before the cosmic / ramp kernels
  hipStream_t side[kStreams] = {nullptr, nullptr};
  hipEvent_t ev_fork[kStreams] = {nullptr, nullptr}, ev_join[kStreams] = {nullptr, nullptr};
  bool fork_narrow = true;
  // Delivered pipeline (fetch_async): the kernels of the exposure that follows on the other stream wait for the
  // KERNELS of the fetched one (not for its copy).  Left alone, the two streams drift into phase -- both compute,
  // then both copy and share the PCIe link: 640 exposures/s -- instead of one copying while the other computes (810).
  hipEvent_t ev_kdone[kStreams] = {nullptr, nullptr};
  bool kdone_valid[kStreams] = {false, false};
  int n_streams = kStreams;              // knob `streams` = 1 serialises all exposures on one stream
  Knobs knobs;                           // frozen at creation (see Knobs)
  DevBuf status_all;                     // kSlots status blocks of kMiscBytes (Slot::misc views into it): ONE copy brings all of them back
  char* status_host = nullptr;           // its pinned host mirror (settle)
  std::string err;
  // grism
  bool have_grism = false;
  GrismDev g{};
  DevBuf sens_wl, sens_val;
  // what the upload keeps per spectrum (host_plan.h): per-bin count rates through this grism's sensitivity, the wide
  // fraction and sigma_l of a bin, the largest PSF sigma and the wavelength range -- cached on the spectrum's content
  plan::SpectrumEstimate est;
  // calibration
  bool have_cal = false;
  int subarray = 0, N = 0, S = 0, cal_R = 0;
  DevBuf flat[4], pfl, sky, lin[4], dark_sci, dark_err, zero_read;
  bool has_flat = false, has_pfl = false, has_sky = false, has_lin = false, has_dark = false,
       has_zero = false;
  float sky_max = 0.f, sky_min = 0.f;                            // range of the positive master sky pixels
  std::vector<float> sky_sorted;                                 // those pixels in ascending order (levels = quantiles)
  // host time of wayne_exposure_upload by part, microseconds (printed at destroy when WAYNE_UPLOAD_TIMING is set)
  double up_us[6] = {0, 0, 0, 0, 0, 0};
  long up_calls = 0;
  Slot slots[kSlots];
  // psf_apply scratch
  DevBuf pa_prefix, pa_nwide, pa_nsplit, pa_nlane, pa_x, pa_y, pa_sl, pa_sh, pa_sub, pa_frame;
  DevBuf pa_in;                 // device mirror of the staging arena (the arrays above are views into it)
  char* pa_stage = nullptr;     // pinned: every input array of a call, copied to the device in one piece
  size_t pa_stage_cap = 0;
  // profiling
  bool prof_on = false;
  unsigned prof_mask = ~0u;   // kernels timed while prof_on (wayne_profile_select)
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  uint64_t prof_launches[WAYNE_PROF_KERNELS] = {0};
  double prof_ms[WAYNE_PROF_KERNELS] = {0};
  uint64_t electrons = 0;  // thrown through wayne_psf_apply (host-counted)
  uint64_t reruns = 0;     // exposures run a second time because a bin lay beyond what the first launch sequence handles
  DevBuf counters;         // kCounterStripes u64 words, 128 B apart: electrons thrown by exposures (device-counted,
                           // count_electrons); word [1]: a spare for debug_fetch's own k_prep_sub launch
};

namespace {

int fail(wayne_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail((ctx), WAYNE_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
  } while (0)

hipEvent_t get_event(wayne_ctx* c) {
  if (!c->ev_pool.empty()) {
    hipEvent_t e = c->ev_pool.back();
    c->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

struct ProfScope {
  wayne_ctx* c;
  ProfRec rec{};
  bool on;
  ProfScope(wayne_ctx* c_, int kernel) : c(c_), on(c_->prof_on && ((c_->prof_mask >> kernel) & 1u)) {
    if (!on) return;
    rec.kernel = kernel;
    rec.a = get_event(c);
    rec.b = get_event(c);
    if (!rec.a || !rec.b) { on = false; return; }
    (void)hipEventRecord(rec.a, c->stream);
  }
  // For a single launch through hipExtLaunchKernel: the events then carry the kernel's own start and stop
  // times (no marker packets before and after it in the stream).
  ProfScope(wayne_ctx* c_, int kernel, bool ext_launch) : c(c_), on(c_->prof_on && ((c_->prof_mask >> kernel) & 1u)) {
    if (!on) return;
    rec.kernel = kernel;
    rec.a = get_event(c);
    rec.b = get_event(c);
    if (!rec.a || !rec.b) { on = false; return; }
    ext = ext_launch;
    if (!ext) (void)hipEventRecord(rec.a, c->stream);
  }
  bool ext = false;
  ~ProfScope() {
    if (!on) return;
    if (!ext) (void)hipEventRecord(rec.b, c->stream);
    c->prof.push_back(rec);
  }
};

int sync_all(wayne_ctx* c) {
  for (int i = 0; i < kStreams; ++i) {
    HIP_TRY(c, hipStreamSynchronize(c->streams[i]));
    if (c->side[i]) HIP_TRY(c, hipStreamSynchronize(c->side[i]));
  }
  return WAYNE_OK;
}

// slot-based calls work on the slot's stream; everything else on stream 0
void use_slot_stream(wayne_ctx* c, int slot) { c->stream = c->streams[slot % c->n_streams]; }

int collect_profile(wayne_ctx* c) {
  int rc = sync_all(c);
  if (rc) return rc;
  for (ProfRec& r : c->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->prof_ms[r.kernel] += ms;
      c->prof_launches[r.kernel] += 1;
    }
    c->ev_pool.push_back(r.a);
    c->ev_pool.push_back(r.b);
  }
  c->prof.clear();
  return WAYNE_OK;
}

template <class T>
int upload(wayne_ctx* c, DevBuf& b, const T* src, size_t n) {
  HIP_TRY(c, b.reserve(std::max<size_t>(n, 1) * sizeof(T)));
  if (n) HIP_TRY(c, hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return WAYNE_OK;
}

inline size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

// (the launch planner -- spectrum estimates, accumulator boxes, sky levels and alias tables -- is host_plan.h: host-only
// code that the CPU harness under tests/native builds with sanitizers)
using plan::build_sky_alias;

// Copy `n` elements into the slot's pinned arena and enqueue the host-to-device copy from there.
template <class T>
int upload_staged(wayne_ctx* c, Slot& s, DevBuf& b, const T* src, size_t n) {
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  if (s.stage_used + bytes > s.stage_cap) return fail(c, WAYNE_E_NOMEM, "upload: staging arena too small");
  if (n) std::memcpy(s.stage + s.stage_used, src, n * sizeof(T));
  b.view((char*)s.in_dev.p + s.stage_used);      // same offset in the device mirror (copied in one piece)
  s.stage_used += align64(bytes);
  return WAYNE_OK;
}

// Plan the sky draws of an exposure (plan::plan_sky) and upload the alias tables it names on the slot's stream with
// the descriptor -- long before k_ramp needs them.
int prepare_sky_tables(wayne_ctx* c, Slot& s) {
  const wayne_exposure_desc& d = s.d;
  plan::SkyPlan sp;
  plan::plan_sky(d.sky_ct_s, s.R, s.read_dt_host.data(), c->has_sky, c->sky_min, c->sky_max, c->sky_sorted, &sp);
  s.sky_pieces = sp.pieces; s.sky_alias_on = false; s.sky_mask = 0; s.sky_L = 1;
  for (int l = 0; l < 16; ++l) { s.sky_level[l] = sp.level[l]; s.sky_tab0[l] = sp.tab0[l]; }
  if (!sp.alias_on) { s.sky_pieces = false; return WAYNE_OK; }    // (no sky, or a read whose rates fit no table: the direct sampler)
  if (sp.keys != s.sky_tab_keys || !s.sky_tab.p) {
    const size_t bytes = (size_t)kMaxReads * kSkyAlias * sizeof(uint32_t);
    HIP_TRY(c, s.sky_tab.reserve(bytes));
    if (!s.sky_tab_host && hipHostMalloc((void**)&s.sky_tab_host, bytes, hipHostMallocDefault) != hipSuccess)
      return fail(c, WAYNE_E_NOMEM, "upload: pinned allocation for the sky tables failed");
    if (!s.sky_tab_ev) HIP_TRY(c, hipEventCreateWithFlags(&s.sky_tab_ev, hipEventDisableTiming));
    if (s.sky_tab_pending) { HIP_TRY(c, hipEventSynchronize(s.sky_tab_ev)); s.sky_tab_pending = false; }
    std::memset(s.sky_tab_host, 0, bytes);
    for (size_t t = 0; t < sp.keys.size() && t < (size_t)kMaxReads; ++t) {
      // (a table is ~1 us to build -- no cache: the sky level, and with it every rate, changes with the exposure)
      float lam;
      std::memcpy(&lam, &sp.keys[t], 4);
      build_sky_alias((double)lam, s.sky_tab_host + t * kSkyAlias);
    }
    HIP_TRY(c, hipMemcpyAsync(s.sky_tab.p, s.sky_tab_host, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(s.sky_tab_ev, c->stream));
    s.sky_tab_pending = true;
    s.sky_tab_keys = sp.keys;
  }
  s.sky_alias_on = true;
  s.sky_mask = sp.mask;
  s.sky_L = sp.L;
  return WAYNE_OK;
}

int side_of(int subarray) { return subarray == 1024 ? 1014 : subarray; }  // detector.py:116-119

// N*N plane -> S*S bordered layout (value `fill` on the 5-px border)
std::vector<float> embed(const float* src, int N, int S, float fill) {
  std::vector<float> v((size_t)S * S, fill);
  for (int y = 0; y < N; ++y)
    std::memcpy(&v[(size_t)(y + kBorder) * S + kBorder], src + (size_t)y * N, (size_t)N * sizeof(float));
  return v;
}

// LDS ints of a thrower workgroup's tile.  A workgroup covers 1/splits of a sub-sample's electrons,
// i.e. a slice of the trace (first-order spectra are < ~260 px long) plus the PSF margin on each
// side, by (margin on each side + the trace's small tilt) rows.  Knob `tile_ints` overrides.
int thrower_lds_ints(const wayne_ctx* c, int splits, int margin) {
  if (c->knobs.tile_ints >= 0) return (int)std::min<long long>(std::max<long long>(c->knobs.tile_ints, 256), 40000);
  const long long w = 260 / std::max(splits, 1) + 2 * margin + 4, h = 2 * margin + 12;
  return (int)std::min<long long>(std::max<long long>(w * h, 1024), 12288);   // <= 48 KiB
}

template <int RNG, int FLUSH>
int launch_throw(wayne_ctx* c, const ThrowArgs& a, int lds_ints) {
  const int groups = (a.K + 7) / 8;
  const dim3 grid((unsigned)(a.K < 8 ? a.K * a.splits : groups * 8 * a.splits));   // (k_throw: block -> (sub-sample, split))
  const size_t lds = (size_t)lds_ints * sizeof(int);
  HIP_TRY(c, hipFuncSetAttribute((const void*)k_throw<RNG, FLUSH>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_throw<RNG, FLUSH>), grid, dim3(kThrowThreads), lds, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return WAYNE_OK;
}

template <int FLUSH>
int launch_narrow(wayne_ctx* c, const ThrowArgs& a, bool exact) {
  const dim3 grid((unsigned)a.K, (unsigned)((a.W + kNarrowThreads - 1) / kNarrowThreads));
  if (exact) hipLaunchKernelGGL((k_narrow<FLUSH, false>), grid, dim3(kNarrowThreads), 0, c->stream, a);
  else hipLaunchKernelGGL((k_narrow<FLUSH, true>), grid, dim3(kNarrowThreads), 0, c->stream, a);
  HIP_TRY(c, hipGetLastError());
  return WAYNE_OK;
}

template <class OutT, bool FAST, int SKY, bool NOISE>
void (*ramp_kernel())(RampArgs) {
  // (pinned to 8 waves per SIMD where that does not spill: see k_ramp / k_ramp_wide)
  if constexpr (FAST && SKY != 0 && !NOISE) return k_ramp<OutT, FAST, SKY, NOISE>;
  else return k_ramp_wide<OutT, FAST, SKY, NOISE>;
}
template <class OutT, bool FAST, int SKY>
void (*ramp_noise(bool noise))(RampArgs) { return noise ? ramp_kernel<OutT, FAST, SKY, true>() : ramp_kernel<OutT, FAST, SKY, false>(); }
template <class OutT, bool FAST>
void (*ramp_sky(int sky, bool noise))(RampArgs) {
  return sky == 1 ? ramp_noise<OutT, FAST, 1>(noise) : sky == 2 ? ramp_noise<OutT, FAST, 2>(noise) : ramp_noise<OutT, FAST, 0>(noise);
}
// k_ramp<reads' type, production / exact math, sky sampler, gaussian-noise stage>
void (*pick_ramp(bool f64, bool exact, int sky, bool noise))(RampArgs) {
  return f64 ? (exact ? ramp_sky<double, false>(sky, noise) : ramp_sky<double, true>(sky, noise))
             : (exact ? ramp_sky<float, false>(sky, noise) : ramp_sky<float, true>(sky, noise));
}

template <int FLUSH>
int launch_lane(wayne_ctx* c, const ThrowArgs& a, bool thin, const PrepArgs* fused_prep = nullptr, const CosmicArgs* fused_cosmic = nullptr) {
  const dim3 grid((unsigned)((a.K + a.kb - 1) / a.kb), (unsigned)((a.W + kLaneThreads - 1) / kLaneThreads));
  if (fused_prep) {
    hipLaunchKernelGGL((k_lane_fused<FLUSH>), grid, dim3(kLaneThreads), 0, c->stream, a, *fused_prep, *fused_cosmic);
  } else if (a.kb > 1) {
    if (thin) hipLaunchKernelGGL((k_lane<FLUSH, true, true>), grid, dim3(kLaneThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((k_lane<FLUSH, false, true>), grid, dim3(kLaneThreads), 0, c->stream, a);
  } else {
    if (thin) hipLaunchKernelGGL((k_lane<FLUSH, true, false>), grid, dim3(kLaneThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((k_lane<FLUSH, false, false>), grid, dim3(kLaneThreads), 0, c->stream, a);
  }
  HIP_TRY(c, hipGetLastError());
  return WAYNE_OK;
}


// The k_ramp instantiation the back half of slot `s` launches, and (if asked) its name as the kernel trace prints it.
void (*select_ramp(const wayne_ctx* c, const Slot& s, std::string* name))(RampArgs) {
  const wayne_exposure_desc& d = s.d;
  const bool f64 = (d.flags & WAYNE_F_OUT_F64) != 0, exact = (d.flags & WAYNE_F_EXACT_SAMPLERS) != 0;
  const int sky_mode = !s.sky_alias_on ? 0 : (s.sky_pieces ? 2 : 1);
  const bool noise = d.noise_mean != 0. && d.noise_std != 0.;
  void (*kern)(RampArgs) = pick_ramp(f64, exact, sky_mode, noise);
  // the production variant with every detector switch on (the rule) has an instantiation of its own (k_ramp.h, ALLON)
  const uint32_t all_on = WAYNE_F_ADD_DARK | WAYNE_F_ADD_NON_LINEAR | WAYNE_F_CLIP_DET_LIMITS | WAYNE_F_ADD_READ_NOISE;
  const bool allon = !f64 && !exact && sky_mode == 1 && !noise && (d.flags & all_on) == all_on && c->has_dark && c->has_lin;
  if (allon) kern = k_ramp<float, true, 1, false, true>;
  if (name) {
    const bool pinned = !exact && sky_mode != 0 && !noise;      // ramp_kernel(): k_ramp where it fits 64 registers, else k_ramp_wide
    char buf[96];
    std::snprintf(buf, sizeof buf, "%s<%s, %s, %d, %s%s>", pinned ? "k_ramp" : "k_ramp_wide", f64 ? "double" : "float",
                  exact ? "false" : "true", sky_mode, noise ? "true" : "false", pinned ? (allon ? ", true" : ", false") : "");
    *name = buf;
  }
  return kern;
}

}  // namespace

extern "C" {

int wayne_abi_version(void) { return WAYNE_ABI_VERSION; }

const char* wayne_strerror(int s) {
  switch (s) {
    case WAYNE_OK: return "ok";
    case WAYNE_E_INVALID: return "invalid argument";
    case WAYNE_E_NEGATIVE: return "negative electron count";
    case WAYNE_E_OVERFLOW: return "electron count overflows the reference's int arithmetic";
    case WAYNE_E_NOMEM: return "out of memory";
    case WAYNE_E_HIP: return "HIP runtime error";
    case WAYNE_E_NODEVICE: return "no gfx950 device";
    case WAYNE_E_STATE: return "grism / calibration / upload missing";
    default: return "unknown status";
  }
}

int wayne_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

wayne_ctx* wayne_ctx_create(int device, int* status) {
  auto set = [&](int s) { if (status) *status = s; };
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
    set(WAYNE_E_NODEVICE);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { set(WAYNE_E_HIP); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set(WAYNE_E_HIP); return nullptr; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    // the code object holds gfx950 ISA only; there is no other path
    set(WAYNE_E_NODEVICE);
    return nullptr;
  }
  wayne_ctx* c = new (std::nothrow) wayne_ctx();
  if (!c) { set(WAYNE_E_NOMEM); return nullptr; }
  c->device = device;
  for (int i = 0; i < kStreams; ++i)
    if (hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipStreamDestroy(c->streams[j]);
      delete c;
      set(WAYNE_E_HIP);
      return nullptr;
    }
  c->stream = c->streams[0];
  for (int i = 0; i < kStreams; ++i) {
    if (hipEventCreateWithFlags(&c->ev_kdone[i], hipEventDisableTiming) != hipSuccess) c->ev_kdone[i] = nullptr;
    if (hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess)
      c->fork_narrow = false;     // not fatal: k_narrow then follows k_throw on the main stream
  }
  // k_narrow beside k_lane on a side stream: worth 10 % when the thrower still had idle issue slots
  // (round 1); now each of the two fills the VALU by itself and running them one after the other is as fast
  // (1860 vs 1846 exposures/s on one stream, 2212 vs 2224 on two) -- off unless WAYNE_FORK_NARROW=1
  // the one place the environment is read (see Knobs)
  for (const KnobName& k : kKnobNames)
    if (const char* e = std::getenv(k.env)) c->knobs.*(k.field) = std::max(std::atoll(e), -1LL);
  c->fork_narrow = c->fork_narrow && c->knobs.fork_narrow > 0;
  if (c->knobs.streams >= 0) c->n_streams = (int)std::min<long long>(std::max<long long>(c->knobs.streams, 1), kStreams);
  if (c->counters.reserve(kCounterBytes) != hipSuccess || hipMemset(c->counters.p, 0, kCounterBytes) != hipSuccess ||
      c->status_all.reserve(kSlots * kMiscBytes) != hipSuccess || hipMemset(c->status_all.p, 0, kSlots * kMiscBytes) != hipSuccess ||
      hipHostMalloc((void**)&c->status_host, kSlots * kMiscBytes, hipHostMallocDefault) != hipSuccess) {
    c->counters.release();
    c->status_all.release();
    if (c->status_host) (void)hipHostFree(c->status_host);
    for (int i = 0; i < kStreams; ++i) (void)hipStreamDestroy(c->streams[i]);
    delete c;
    set(WAYNE_E_NOMEM);
    return nullptr;
  }
  set(WAYNE_OK);
  return c;
}

void wayne_ctx_destroy(wayne_ctx* c) {
  if (!c) return;
  if (c->knobs.upload_timing > 0 && c->up_calls > 0)
    std::fprintf(stderr, "wayne_exposure_upload: %ld calls; us per call: staging + copies %.1f, electron estimate %.1f, "
                 "accumulator boxes %.1f, sky tables %.1f\n", c->up_calls, c->up_us[0] / c->up_calls,
                 c->up_us[1] / c->up_calls, c->up_us[2] / c->up_calls, c->up_us[3] / c->up_calls);
  (void)hipSetDevice(c->device);
  (void)sync_all(c);
  for (ProfRec& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (Slot& s : c->slots) s.release();
  if (c->status_host) (void)hipHostFree(c->status_host);
  for (DevBuf* b : {&c->counters, &c->status_all, &c->sens_wl, &c->sens_val, &c->pfl, &c->sky, &c->dark_sci, &c->dark_err, &c->zero_read,
                    &c->pa_prefix, &c->pa_nwide, &c->pa_nsplit, &c->pa_nlane, &c->pa_x, &c->pa_y, &c->pa_sl, &c->pa_sh, &c->pa_sub,
                    &c->pa_frame, &c->pa_in})
    b->release();
  if (c->pa_stage) (void)hipHostFree(c->pa_stage);
  for (int i = 0; i < 4; ++i) { c->flat[i].release(); c->lin[i].release(); }
  for (int i = 0; i < kStreams; ++i) {
    (void)hipStreamDestroy(c->streams[i]);
    if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    if (c->ev_fork[i]) (void)hipEventDestroy(c->ev_fork[i]);
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    if (c->ev_kdone[i]) (void)hipEventDestroy(c->ev_kdone[i]);
  }
  delete c;
}

const char* wayne_last_error(const wayne_ctx* c) { return c ? c->err.c_str() : "null context"; }

// Every slot whose last whole run nobody has looked at yet: read its status word (all of them in ONE copy) and run a
// slot that met a bin beyond its launch sequence's reach a second time, with the general sequence.  Called with all
// streams idle; leaves them idle.  An overflow (bit 0) is reported after the other slots have been settled.
static int settle(wayne_ctx* c) {
  int err = WAYNE_OK;
  for (int pass = 0; pass < 2; ++pass) {
    int hi = -1;
    for (int i = 0; i < kSlots; ++i)
      if (c->slots[i].ran && c->slots[i].uploaded) hi = i;
    if (hi < 0) break;
    HIP_TRY(c, hipMemcpy(c->status_host, c->status_all.p, (size_t)(hi + 1) * kMiscBytes, hipMemcpyDeviceToHost));
    bool again = false;
    for (int i = 0; i <= hi; ++i) {
      Slot& s = c->slots[i];
      if (!s.ran || !s.uploaded) continue;
      s.ran = false;
      const int status = ((const Slot::Misc*)(c->status_host + (size_t)i * kMiscBytes))->status;
      if (status & 1) {
        err = fail(c, WAYNE_E_OVERFLOW, "exposure: a sub-sample holds >= 2^32 electrons (or a bin >= 2^31)");
      } else if ((status & 2) && pass == 0) {
        s.force_throw = true;
        c->reruns += 1;
        int rc = wayne_exposure_run(c, i);
        if (rc) return rc;
        again = true;
      }
    }
    if (!again) break;
    int rc = sync_all(c);
    if (rc) return rc;
  }
  return err;
}

int wayne_ctx_synchronize(wayne_ctx* c) {
  if (!c) return WAYNE_E_INVALID;
  (void)hipSetDevice(c->device);
  int rc = sync_all(c);
  if (rc) return rc;
  return settle(c);
}

int wayne_ctx_set_knob(wayne_ctx* c, const char* name, long long value) {
  if (!c || !name) return WAYNE_E_INVALID;
  for (const KnobName& k : kKnobNames)
    if (std::strcmp(name, k.name) == 0) {
      if (k.field == &Knobs::streams || k.field == &Knobs::fork_narrow) {
        // (these two shape the context itself: no exposure may be in flight while they change)
        int rc = sync_all(c);
        if (rc) return rc;
      }
      c->knobs.*(k.field) = std::max(value, -1LL);
      if (k.field == &Knobs::streams)
        c->n_streams = value < 0 ? kStreams : (int)std::min<long long>(std::max<long long>(value, 1), kStreams);
      if (k.field == &Knobs::fork_narrow) c->fork_narrow = value > 0 && c->side[0] && c->side[1] && c->ev_fork[0] && c->ev_fork[1] && c->ev_join[0] && c->ev_join[1];
      return WAYNE_OK;
    }
  return fail(c, WAYNE_E_INVALID, std::string("set_knob: no knob named ") + name);
}

int wayne_ctx_get_knob(const wayne_ctx* c, const char* name, long long* value) {
  if (!c || !name || !value) return WAYNE_E_INVALID;
  for (const KnobName& k : kKnobNames)
    if (std::strcmp(name, k.name) == 0) { *value = c->knobs.*(k.field); return WAYNE_OK; }
  return WAYNE_E_INVALID;
}

const char* wayne_build_flags(void) {
  // every compile-time switch that changes what the library computes or launches; the shipped build has none
  return ""
#ifdef WAYNE_NEGCTL_SKY_RUNAWAY
         " WAYNE_NEGCTL_SKY_RUNAWAY"
#endif
#ifdef WAYNE_NEGCTL_ADDITIVE_KEY
         " WAYNE_NEGCTL_ADDITIVE_KEY"
#endif
#ifdef WAYNE_NEGCTL_DROP_FRACTION
         " WAYNE_NEGCTL_DROP_FRACTION"
#endif
#ifdef WAYNE_TIMING_KNOBS
         " WAYNE_TIMING_KNOBS"
#endif
#ifdef WAYNE_TIMING_COLPOOL
         " WAYNE_TIMING_COLPOOL"
#endif
#ifdef WAYNE_TIMING_RAMP_NO_DARK
         " WAYNE_TIMING_RAMP_NO_DARK"
#endif
#ifdef WAYNE_TIMING_RAMP_NO_ONCE
         " WAYNE_TIMING_RAMP_NO_ONCE"
#endif
#if WAYNE_PREP_THREADS != 512
         " WAYNE_PREP_THREADS"
#endif
#if WAYNE_RAMP_THREADS != 1024
         " WAYNE_RAMP_THREADS"
#endif
#if WAYNE_RAMP_PF != 2
         " WAYNE_RAMP_PF"
#endif
      ;
}

void* wayne_ctx_stream(wayne_ctx* c) { return c ? (void*)c->streams[0] : nullptr; }

int wayne_ctx_slots(const wayne_ctx*) { return kSlots; }

// ---------------------------------------------------------------------------
// wayne_psf_apply
// ---------------------------------------------------------------------------
int wayne_psf_apply(wayne_ctx* c, const int32_t* counts, int size, const double* x_pos,
                    const double* y_pos, const double* psf_ratio, const double* psf_sigmal,
                    const double* psf_sigmah, int nr, int nc, uint32_t seed, int threads_compat,
                    int rng_mode, uint32_t exposure, uint32_t subsample, int32_t* out) {
  return wayne_psf_apply_ex(c, counts, size, x_pos, y_pos, psf_ratio, psf_sigmal, psf_sigmah, nr, nc, seed, threads_compat,
                            rng_mode, exposure, subsample, 0u, out);
}

int wayne_psf_apply_ex(wayne_ctx* c, const int32_t* counts, int size, const double* x_pos,
                       const double* y_pos, const double* psf_ratio, const double* psf_sigmal,
                       const double* psf_sigmah, int nr, int nc, uint32_t seed, int threads_compat,
                       int rng_mode, uint32_t exposure, uint32_t subsample, uint32_t flags, int32_t* out) {
  if (!c) return WAYNE_E_INVALID;
  if (size < 0 || nr <= 0 || nc <= 0 || !out) return fail(c, WAYNE_E_INVALID, "psf_apply: bad size / frame");
  if (nr != nc)
    return fail(c, WAYNE_E_INVALID,
                "psf_apply: only square frames (the reference indexes ypos*nc+xpos but bounds xpos by nr)");
  if (size > 0 && (!counts || !x_pos || !y_pos || !psf_ratio || !psf_sigmal || !psf_sigmah))
    return fail(c, WAYNE_E_INVALID, "psf_apply: null input");
  if (rng_mode != WAYNE_RNG_REPLAY && rng_mode != WAYNE_RNG_PHILOX && rng_mode != WAYNE_RNG_SPLIT)
    return fail(c, WAYNE_E_INVALID, "psf_apply: rng_mode");
  if (rng_mode == WAYNE_RNG_REPLAY && threads_compat <= 0)
    return fail(c, WAYNE_E_INVALID, "psf_apply: threads_compat must be >= 1 in replay mode");
  const int N = nr;
  (void)hipSetDevice(c->device);
  c->stream = c->streams[0];

  // A1 (ssum, with the reference's silent int overflows turned into errors), N = (int)(counts * ratio), the split mode's
  // routing and the tiles' clip rectangle: plan::plan_psf_apply (host_plan.h: host-only code, under sanitizers in tests/native)
  const int margin = 30;
  plan::PsfPlan pp;
  switch (plan::plan_psf_apply(counts, size, x_pos, y_pos, psf_ratio, psf_sigmal, N, rng_mode, threads_compat, margin, &pp)) {
    case plan::PSF_NEGATIVE: return fail(c, WAYNE_E_NEGATIVE, "psf_apply: negative count");
    case plan::PSF_OVERFLOW_REPLAY: return fail(c, WAYNE_E_OVERFLOW, "psf_apply: sum(counts)*threads >= 2^31 (pyparallel_menu.c:12,48)");
    case plan::PSF_OVERFLOW_TOTAL: return fail(c, WAYNE_E_OVERFLOW, "psf_apply: more than 2^32-1 electrons");
    default: break;
  }
  const long long total = pp.total;

  HIP_TRY(c, c->pa_frame.reserve((size_t)N * N * sizeof(int32_t)));
  HIP_TRY(c, hipMemsetAsync(c->pa_frame.p, 0, (size_t)N * N * sizeof(int32_t), c->stream));  // A3

  if (total > 0) {
    const std::vector<uint32_t>& prefix = pp.prefix;
    const std::vector<int32_t>&nwide = pp.nwide, &nsplit = pp.nsplit, &nlane = pp.nlane;
    const uint32_t run = pp.run;
    const bool any_split = pp.any_split, any_lane = pp.any_lane;
    SubInfo si{};
    si.electrons = run;
    si.read = 0;
    si.replay_seed = (int)seed;
    si.tx0 = pp.tx0; si.ty0 = pp.ty0; si.tw = pp.tw; si.th = pp.th;   // clip region of the tiles
    int rc;
    {
      // one pinned arena, one host-to-device copy (nine pageable copies cost ~0.1 ms of a 0.35 ms call)
      const size_t n = (size_t)size;
      const size_t need = align64((n + 1) * 4) + 3 * align64(n * 4) + 4 * align64(n * 8) + align64(sizeof(SubInfo)) + 64;
      if (c->pa_stage_cap < need) {
        if (c->pa_stage) (void)hipHostFree(c->pa_stage);
        c->pa_stage = nullptr;
        c->pa_stage_cap = 0;
        if (hipHostMalloc((void**)&c->pa_stage, need, hipHostMallocDefault) != hipSuccess)
          return fail(c, WAYNE_E_NOMEM, "psf_apply: pinned staging allocation failed");
        c->pa_stage_cap = need;
      }
      HIP_TRY(c, c->pa_in.reserve(c->pa_stage_cap));
      size_t used = 0;
      auto stage = [&](DevBuf& b, const void* src, size_t bytes) {
        if (bytes) std::memcpy(c->pa_stage + used, src, bytes);
        b.view((char*)c->pa_in.p + used);
        used += align64(std::max<size_t>(bytes, 1));
      };
      stage(c->pa_prefix, prefix.data(), (n + 1) * 4);
      stage(c->pa_nwide, nwide.data(), n * 4);
      stage(c->pa_nsplit, nsplit.data(), n * 4);
      stage(c->pa_nlane, nlane.data(), n * 4);
      stage(c->pa_x, x_pos, n * 8);
      stage(c->pa_y, y_pos, n * 8);
      stage(c->pa_sl, psf_sigmal, n * 8);
      stage(c->pa_sh, psf_sigmah, n * 8);
      stage(c->pa_sub, &si, sizeof si);
      HIP_TRY(c, hipMemcpyAsync(c->pa_in.p, c->pa_stage, used, hipMemcpyHostToDevice, c->stream));
    }

    ThrowArgs a{};
    a.W = size; a.K = 1; a.N = N; a.S = N + 2 * kBorder; a.kb = 1;
    // enough workgroups to fill the chip when the call is big, one when small
    a.splits = (int)std::min<long long>(512, std::max<long long>(1, total / (64LL * kThrowThreads)));
    a.min_wgs = a.splits;   // one call, one sub-sample: share the electrons among all launched workgroups
    a.threads_compat = threads_compat;
    a.margin = margin;
    const int lds_ints = thrower_lds_ints(c, a.splits, margin);
    a.lds_ints = lds_ints;
    a.seed = seed; a.exposure = exposure; a.subsample0 = subsample;
    a.flags = 0; a.flat_off = 0; a.flat_wmin = 0; a.flat_wmax = 1; a.flat_inv_range = 1;
    a.sub = c->pa_sub.as<SubInfo>();
    a.prefix = c->pa_prefix.as<uint32_t>();
    a.nwide = c->pa_nwide.as<int32_t>();
    a.nsplit = c->pa_nsplit.as<int32_t>();
    a.nlane = c->pa_nlane.as<int32_t>();
    a.xpos = c->pa_x.as<double>(); a.ypos = c->pa_y.as<double>();
    a.sigl = c->pa_sl.as<double>(); a.sigh = c->pa_sh.as<double>();
    for (int i = 0; i < 4; ++i) a.flat[i] = nullptr;
    a.acc = nullptr;
    a.frame = c->pa_frame.as<int32_t>();
    if ((size + kNarrowThreads - 1) / kNarrowThreads > kMaxChunks && (any_lane || any_split))
      return fail(c, WAYNE_E_INVALID, "psf_apply: more than 65536 bins in split mode");
    for (int i = 0; i < kMaxChunks; ++i) a.chunk_order[i] = a.lane_order[i] = (unsigned char)i;
    if (run > 0) {
      ProfScope ps(c, PK_THROW);
      rc = (rng_mode == WAYNE_RNG_REPLAY) ? launch_throw<0, 0>(c, a, lds_ints) : launch_throw<1, 0>(c, a, lds_ints);
      if (rc) return rc;
    }
    if (any_lane) {
      ProfScope ps(c, PK_LANE);
      if ((rc = launch_lane<0>(c, a, false))) return rc;
    }
    if (any_split) {
      ProfScope ps(c, PK_NARROW);
      if ((rc = launch_narrow<0>(c, a, (flags & WAYNE_F_EXACT_SAMPLERS) != 0))) return rc;
    }
    c->electrons += (uint64_t)total;
  }
  // (the frame goes straight into the caller's pageable array: landing it in a pinned buffer first and copying it
  // from there was 0.1 ms slower per call)
  HIP_TRY(c, hipMemcpyAsync(out, c->pa_frame.p, (size_t)N * N * sizeof(int32_t), hipMemcpyDeviceToHost,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return WAYNE_OK;
}

// ---------------------------------------------------------------------------
// grism + calibration
// ---------------------------------------------------------------------------
int wayne_ctx_set_grism(wayne_ctx* c, const wayne_grism_desc* g) {
  if (!c || !g) return WAYNE_E_INVALID;
  if (g->n_sens < 0 || (g->n_sens > 0 && (!g->sens_wl_um || !g->sens_val)))
    return fail(c, WAYNE_E_INVALID, "set_grism: sensitivity table");
  if (!plan::sens_table_ok(g->sens_wl_um, g->sens_val, g->n_sens))
    return fail(c, WAYNE_E_INVALID, "set_grism: sensitivity table must have finite values at finite, non-decreasing wavelengths");
  (void)hipSetDevice(c->device);
  c->stream = c->streams[0];
  int rc;
  if ((rc = upload(c, c->sens_wl, g->sens_wl_um, (size_t)g->n_sens))) return rc;
  if ((rc = upload(c, c->sens_val, g->sens_val, (size_t)g->n_sens))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  GrismDev& d = c->g;
  std::memcpy(d.trace, g->trace_coeff, sizeof d.trace);
  std::memcpy(d.wlsol, g->wl_solution, sizeof d.wlsol);
  std::memcpy(d.p_ratio, g->psf_ratio_poly, sizeof d.p_ratio);
  std::memcpy(d.p_sigl, g->psf_sigmal_poly, sizeof d.p_sigl);
  std::memcpy(d.p_sigh, g->psf_sigmah_poly, sizeof d.p_sigh);
  d.flat_wmin = g->flat_wmin;
  d.flat_wmax = g->flat_wmax;
  d.n_sens = g->n_sens;
  d.sens_wl = c->sens_wl.as<double>();
  d.sens_val = c->sens_val.as<double>();
  c->have_grism = true;
  // (... which also forgets what the upload kept per spectrum: it was worked out with the previous grism)
  c->est.set_grism(d, g->sens_wl_um, g->sens_val, g->n_sens);
  return WAYNE_OK;
}

int wayne_ctx_set_calibration(wayne_ctx* c, const wayne_calibration* k) {
  if (!c || !k) return WAYNE_E_INVALID;
  const int sub = k->subarray;
  if (sub != 64 && sub != 128 && sub != 256 && sub != 512 && sub != 1024)
    return fail(c, WAYNE_E_INVALID, "set_calibration: SUBARRAY must be 64,128,256,512 or 1024");
  if (k->n_reads < 1 || k->n_reads > kMaxReads) return fail(c, WAYNE_E_INVALID, "set_calibration: n_reads must be 1..15");
  (void)hipSetDevice(c->device);
  { int rc0 = sync_all(c); if (rc0) return rc0; }   // no exposure may be in flight while planes change
  c->stream = c->streams[0];
  const int N = side_of(sub), S = N + 2 * kBorder;
  const size_t NN = (size_t)N * N, SS = (size_t)S * S;
  int rc;
  c->has_flat = k->flat[0] && k->flat[1] && k->flat[2] && k->flat[3];
  if (c->has_flat)
    for (int i = 0; i < 4; ++i)
      if ((rc = upload(c, c->flat[i], k->flat[i], NN))) return rc;
  c->has_pfl = k->pfl != nullptr;
  if (c->has_pfl) {
    std::vector<float> v = embed(k->pfl, N, S, 1.0f);
    if ((rc = upload(c, c->pfl, v.data(), SS))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->has_sky = k->sky != nullptr;
  c->sky_max = 0.f;
  c->sky_min = 0.f;
  if (c->has_sky) {
    c->sky_sorted.clear();
    for (size_t i = 0; i < NN; ++i)
      if (k->sky[i] > 0.f) c->sky_sorted.push_back(k->sky[i]);
    std::sort(c->sky_sorted.begin(), c->sky_sorted.end());
    if (!c->sky_sorted.empty()) { c->sky_min = c->sky_sorted.front(); c->sky_max = c->sky_sorted.back(); }
    std::vector<float> v = embed(k->sky, N, S, 0.0f);
    if ((rc = upload(c, c->sky, v.data(), SS))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  c->has_lin = k->lin[0] && k->lin[1] && k->lin[2] && k->lin[3];
  if (c->has_lin)
    for (int i = 0; i < 4; ++i)
      if ((rc = upload(c, c->lin[i], k->lin[i], SS))) return rc;
  c->has_dark = k->dark_sci && k->dark_err;
  if (c->has_dark) {
    if ((rc = upload(c, c->dark_sci, k->dark_sci, SS * k->n_reads))) return rc;
    {
      // `err <= 0 -> 1e-5` (detector.py:189-190) once, here, instead of a compare and a select per pixel and read in
      // k_ramp: the plane in HBM already holds what the reference's np.where would hand to np.random.normal
      std::vector<float> e(k->dark_err, k->dark_err + SS * k->n_reads);
      for (float& x : e) if (!(x > 0.f)) x = 0.00001f;
      if ((rc = upload(c, c->dark_err, e.data(), e.size()))) return rc;
      HIP_TRY(c, hipStreamSynchronize(c->stream));      // (the copy reads a local vector)
    }
  }
  c->has_zero = k->zero_read != nullptr;
  if (c->has_zero)
    if ((rc = upload(c, c->zero_read, k->zero_read, SS))) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->subarray = sub; c->N = N; c->S = S; c->cal_R = k->n_reads;
  c->have_cal = true;
  for (Slot& s : c->slots) { s.uploaded = false; s.acc_init = false; }
  return WAYNE_OK;
}

// ---------------------------------------------------------------------------
// exposure
// ---------------------------------------------------------------------------
int wayne_exposure_upload(wayne_ctx* c, int slot, const wayne_exposure_desc* d) {
  if (!c || !d) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "upload: slot");
  if (!c->have_grism || !c->have_cal) return fail(c, WAYNE_E_STATE, "upload: set grism and calibration first");
  const int W = d->n_wl, K = d->n_samples, R = d->n_reads;
  if (W < 2 || K < 1 || R < 1) return fail(c, WAYNE_E_INVALID, "upload: need n_wl >= 2, n_samples >= 1, n_reads >= 1");
  if (R != c->cal_R) return fail(c, WAYNE_E_INVALID, "upload: n_reads differs from the calibration's");
  if (!d->wl_um || !d->flux || !d->x_ref || !d->y_ref || !d->dur_ms || !d->sample_read || !d->read_dt_s)
    return fail(c, WAYNE_E_INVALID, "upload: null array");
  if (d->rng_mode != WAYNE_RNG_REPLAY && d->rng_mode != WAYNE_RNG_PHILOX && d->rng_mode != WAYNE_RNG_SPLIT)
    return fail(c, WAYNE_E_INVALID, "upload: rng_mode");
  if (d->rng_mode == WAYNE_RNG_REPLAY && (d->threads_compat <= 0 || !d->replay_seed))
    return fail(c, WAYNE_E_INVALID, "upload: replay mode needs threads_compat >= 1 and replay_seed");
  for (int k = 0; k < K; ++k)
    if (d->sample_read[k] < 0 || d->sample_read[k] >= R) return fail(c, WAYNE_E_INVALID, "upload: sample_read out of range");
  if ((long long)K * W > 0x7FFFFFFFLL) return fail(c, WAYNE_E_INVALID, "upload: K*W too large");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  Slot& s = c->slots[slot];
  const auto t_up0 = std::chrono::steady_clock::now();
  auto lap = [&](int part, std::chrono::steady_clock::time_point& t_) {
    const auto now = std::chrono::steady_clock::now();
    c->up_us[part] += std::chrono::duration<double, std::micro>(now - t_).count();
    t_ = now;
  };
  auto t_lap = t_up0;
  // the slot is unusable until this call has succeeded: a failure half way leaves re-pointed views behind
  s.uploaded = false;
  s.front_done = false;
  s.fused_last = false;      // (last_prep points into buffers this call may re-allocate)
  int rc;
  const size_t KW = (size_t)K * W;
  {
    // staging arena: every array of the descriptor, 64-byte aligned
    size_t need = 2 * align64((size_t)W * 8) + 3 * align64((size_t)K * 8) + 2 * align64((size_t)K * 4) + align64((size_t)R * 8) + 1024;
    if (d->lc_z) need += 2 * align64((size_t)K * 8) + align64((size_t)W * 8);
    if (d->depth) need += align64(KW * 8);
    if (s.stage_pending) {           // the previous upload of this slot may still be reading the arena
      HIP_TRY(c, hipEventSynchronize(s.stage_ev));
      s.stage_pending = false;
    }
    if (s.stage_cap < need) {
      if (s.stage) (void)hipHostFree(s.stage);
      s.stage = nullptr;
      s.stage_cap = 0;
      if (hipHostMalloc((void**)&s.stage, need, hipHostMallocDefault) != hipSuccess)
        return fail(c, WAYNE_E_NOMEM, "upload: pinned staging allocation failed");
      s.stage_cap = need;
    }
    if (!s.stage_ev) HIP_TRY(c, hipEventCreateWithFlags(&s.stage_ev, hipEventDisableTiming));
    s.stage_used = 0;
    HIP_TRY(c, s.in_dev.reserve(s.stage_cap));
  }
  if ((rc = upload_staged(c, s, s.wl, d->wl_um, (size_t)W))) return rc;
  if ((rc = upload_staged(c, s, s.flux, d->flux, (size_t)W))) return rc;
  s.has_lc = d->lc_z != nullptr;
  if (s.has_lc) {
    if (d->depth) return fail(c, WAYNE_E_INVALID, "upload: give either depth or lc_z, not both");
    if (!d->lc_rp) return fail(c, WAYNE_E_INVALID, "upload: lc_z needs lc_rp");
    if ((rc = upload_staged(c, s, s.lc_z, d->lc_z, (size_t)K))) return rc;
    if ((rc = upload_staged(c, s, s.lc_rp, d->lc_rp, (size_t)W))) return rc;
    s.lc_p_lo = s.lc_p_hi = d->lc_rp[0];
    for (int i = 1; i < W; ++i) { s.lc_p_lo = std::min(s.lc_p_lo, d->lc_rp[i]); s.lc_p_hi = std::max(s.lc_p_hi, d->lc_rp[i]); }
    s.has_lc_hidden = d->lc_hidden != nullptr;
    if (s.has_lc_hidden && (rc = upload_staged(c, s, s.lc_hidden, d->lc_hidden, (size_t)K))) return rc;
    HIP_TRY(c, s.depth.reserve(KW * sizeof(double)));
  }
  s.has_depth = d->depth != nullptr || s.has_lc;
  if (d->depth && (rc = upload_staged(c, s, s.depth, d->depth, KW))) return rc;
  if ((rc = upload_staged(c, s, s.xref, d->x_ref, (size_t)K))) return rc;
  if ((rc = upload_staged(c, s, s.yref, d->y_ref, (size_t)K))) return rc;
  if ((rc = upload_staged(c, s, s.dur, d->dur_ms, (size_t)K))) return rc;
  s.has_replay_seed = d->replay_seed != nullptr;
  if (s.has_replay_seed && (rc = upload_staged(c, s, s.rseed, d->replay_seed, (size_t)K))) return rc;
  if ((rc = upload_staged(c, s, s.sread, d->sample_read, (size_t)K))) return rc;
  if ((rc = upload_staged(c, s, s.read_dt, d->read_dt_s, (size_t)R))) return rc;
  HIP_TRY(c, hipMemcpyAsync(s.in_dev.p, s.stage, s.stage_used, hipMemcpyHostToDevice, c->stream));   // one copy for all arrays
  HIP_TRY(c, hipEventRecord(s.stage_ev, c->stream));
  s.stage_pending = true;
  for (DevBuf* b : {&s.ratio, &s.sigl, &s.sigh, &s.sens, &s.dlam}) HIP_TRY(c, b->reserve((size_t)W * sizeof(double)));
  HIP_TRY(c, s.counts.reserve(KW * sizeof(int32_t)));
  HIP_TRY(c, s.nwide.reserve(KW * sizeof(int32_t)));
  HIP_TRY(c, s.nsplit.reserve(KW * sizeof(int32_t)));
  HIP_TRY(c, s.nlane.reserve(KW * sizeof(int32_t)));
  HIP_TRY(c, s.prefix.reserve((size_t)K * (W + 1) * sizeof(uint32_t)));
  HIP_TRY(c, s.xpos.reserve(KW * sizeof(double)));
  HIP_TRY(c, s.ypos.reserve(KW * sizeof(double)));
  HIP_TRY(c, s.sub.reserve((size_t)K * sizeof(SubInfo)));
  HIP_TRY(c, s.tr.reserve((size_t)K * kTrStride * sizeof(double)));
  {
    const size_t n_chunks = (size_t)(W + kPrepThreads - 1) / kPrepThreads;
    if (n_chunks > (size_t)kMaxPrepChunks || W > 32768) return fail(c, WAYNE_E_INVALID, "upload: more than 32768 wavelength bins");
    HIP_TRY(c, s.chunk_total.reserve((size_t)K * n_chunks * sizeof(uint32_t)));
    HIP_TRY(c, s.chunk_box.reserve((size_t)K * n_chunks * 4 * sizeof(double)));
  }
  s.misc.view((char*)c->status_all.p + (size_t)slot * kMiscBytes);
  s.ran = false;
  const size_t SS = (size_t)c->S * c->S;
  const size_t acc_bytes = (size_t)R * SS * sizeof(long long);
  if (s.acc.cap < acc_bytes) s.acc_init = false;
  HIP_TRY(c, s.acc.reserve(acc_bytes));
  if (!s.acc_init) {
    HIP_TRY(c, hipMemsetAsync(s.acc.p, 0, acc_bytes, c->stream));
    s.acc_init = true;
    s.acc_dirty = false;
  }
  const size_t out_elem = (d->flags & WAYNE_F_OUT_F64) ? sizeof(double) : sizeof(float);
  HIP_TRY(c, s.out.reserve((size_t)(R + 1) * SS * out_elem));
  // (no stream synchronisation: the host arrays were copied into the pinned arena above)
  s.d = *d;
  s.d.wl_um = s.d.flux = s.d.depth = s.d.x_ref = s.d.y_ref = s.d.dur_ms = s.d.read_dt_s = nullptr;
  s.d.replay_seed = s.d.sample_read = nullptr;
  s.d.lc_z = s.d.lc_hidden = s.d.lc_rp = nullptr;
  s.W = W; s.K = K; s.R = R;
  s.read_dt_host.assign(d->read_dt_s, d->read_dt_s + R);
  lap(0, t_lap);     // staging, copies, reservations
  {
    plan::ThrowPlan tp;
    plan::estimate_thrown(c->est, W, d->wl_um, d->flux, K, d->dur_ms, d->scale_factor, d->rng_mode, &tp);
    s.est_thrown = tp.est_thrown; s.max_chunk_electrons = tp.max_chunk_electrons; s.max_narrow = tp.max_narrow;
    std::memcpy(s.chunk_order, tp.chunk_order, sizeof s.chunk_order);
    std::memcpy(s.lane_order, tp.lane_order, sizeof s.lane_order);
  }
  lap(1, t_lap);
  {
    // k_lane's batches and its first-touch flush list (plan::lane_batches)
    int kb = 1;
    plan::lane_batches(K, W, s.max_chunk_electrons, &kb, &s.thin);
    if (c->knobs.batch >= 0) kb = (int)std::min<long long>(std::max<long long>(c->knobs.batch, 1), kLaneBatchMax);
    s.kb = kb;
    if (c->knobs.thin >= 0) s.thin = c->knobs.thin != 0;
  }
  s.use_box = plan::accumulator_boxes(c->est, W, d->wl_um, d->flux, K, R, c->S, d->sub_scale, d->x_ref, d->y_ref,
                                      d->sample_read, s.acc_box) && !(c->knobs.no_acc_box > 0);
  lap(2, t_lap);
  {
    const size_t seg_bytes = ((SS + 63) / 64) * sizeof(uint32_t);
    if (s.seg.cap < seg_bytes) {
      HIP_TRY(c, s.seg.reserve(seg_bytes));
      HIP_TRY(c, hipMemsetAsync(s.seg.p, 0, s.seg.cap, c->stream));
    }
  }
  if ((rc = prepare_sky_tables(c, s))) return rc;
  lap(3, t_lap);
  c->up_calls += 1;
  s.force_throw = false;
  s.uploaded = true;
  s.front_done = false;
  return WAYNE_OK;
}

int wayne_exposure_run_front(wayne_ctx* c, int slot) {
  if (!c) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "run: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "run: slot not uploaded");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  const wayne_exposure_desc& d = s.d;
  const int W = s.W, K = s.K, R = s.R, N = c->N, S = c->S;
  const size_t SS = (size_t)S * S;
  if (c->n_streams == 2) {
    const int other = 1 - slot % 2;
    if (c->kdone_valid[other]) {     // see wayne_ctx::ev_kdone
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_kdone[other], 0));
      c->kdone_valid[other] = false;
    }
  }
  if (s.acc_dirty) {     // a front half without its back half: start from clean accumulators (and segment bits)
    HIP_TRY(c, hipMemsetAsync(s.acc.p, 0, (size_t)R * SS * sizeof(long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(s.seg.p, 0, s.seg.cap, c->stream));
    s.acc_dirty = false;
  }
  if (s.has_lc) {
    LcArgs a{};
    a.K = K; a.W = W;
    a.z = s.lc_z.as<double>();
    a.hidden = s.has_lc_hidden ? s.lc_hidden.as<double>() : nullptr;
    a.rp = s.lc_rp.as<double>();
    double sum = 0.;
    for (int i = 0; i < 4; ++i) { a.ld[i] = d.lc_ld[i]; sum += d.lc_ld[i] * (i + 1) / (i + 5.0); }
    a.f0 = kPi * (1.0 - sum);
    // tanh-sinh rule on (0, 1): t in [-3, 3], u = pi/2 sinh t, x = (1 + tanh u)/2 (wayne_amd/lightcurve.py)
    const double h = 6.0 / (kLcNodes - 1);
    for (int i = 0; i < kLcNodes; ++i) {
      const double t = -3.0 + h * i, u = 0.5 * kPi * std::sinh(t);
      a.x[i] = (float)(0.5 * (1.0 + std::tanh(u)));
      a.w[i] = (float)(h * 0.25 * kPi * std::cosh(t) / (std::cosh(u) * std::cosh(u)));
      a.d[i] = (float)(0.5 * std::exp(-std::fabs(u)) / std::cosh(u));
    }
    a.p_lo = s.lc_p_lo; a.p_hi = s.lc_p_hi;
    a.depth = s.depth.as<double>();
    ProfScope ps(c, PK_LIGHTCURVE);
    hipLaunchKernelGGL(k_lightcurve, dim3(K), dim3(256), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
  }

  WlArrays wa{s.ratio.as<double>(), s.sigl.as<double>(), s.sigh.as<double>(), s.sens.as<double>(),
              s.dlam.as<double>()};
  {
    ProfScope ps(c, PK_PREP_WL);
    hipLaunchKernelGGL(k_prep_wl, dim3((std::max(W, K) + 255) / 256), dim3(256), 0, c->stream, c->g, W,
                       s.wl.as<double>(), wa, s.misc.as<uint32_t>(), K, s.xref.as<double>(), s.yref.as<double>(),
                       s.tr.as<double>());
    HIP_TRY(c, hipGetLastError());
  }
  // margin of a thrower workgroup's tile around its slice of the trace: 5 sigma_h, so that practically no electron
  // takes the in-loop global-atomic path (see k_lane)
  bool skip_narrow = false;
  bool lane_unlimited = false;   // split mode without a k_throw launch: the lanes take every bin (up to kLaneReach)
  bool fused = false;            // ... and no k_prep_sub either: k_lane<.., FUSED> plans its bins itself
  const int margin = d.thrower_margin > 0 ? d.thrower_margin : 30;
  PrepArgs prep_args{};
  CosmicArgs cosmic_args{};
  {
    PrepArgs& a = prep_args;
    a.g = c->g;
    a.W = W; a.K = K; a.N = N;
    a.sub_scale = d.sub_scale;
    a.margin = margin;
    a.max_tile = 0x7FFFFFFF;      // the sub-sample rectangle is only the clip region of the workgroup tiles
    a.seed = d.seed; a.exposure = d.exposure_index; a.flags = d.flags;
    a.scale_factor = d.scale_factor;
    a.wl = s.wl.as<double>(); a.flux = s.flux.as<double>();
    a.depth = s.has_depth ? s.depth.as<double>() : nullptr;
    a.x_ref = s.xref.as<double>(); a.y_ref = s.yref.as<double>(); a.dur_ms = s.dur.as<double>();
    a.replay_seed = s.has_replay_seed ? s.rseed.as<int32_t>() : nullptr;
    a.sample_read = s.sread.as<int32_t>();
    a.tr = s.tr.as<double>();
    a.wa = wa;
    a.counts = s.counts.as<int32_t>(); a.nwide = s.nwide.as<int32_t>();
    a.nsplit = s.nsplit.as<int32_t>();
    a.nlane = s.nlane.as<int32_t>();
    a.split_min = (d.rng_mode == WAYNE_RNG_SPLIT) ? kSplitMin : 0;
    // no bin expected beyond a lane's cap (the rule on every BASELINE configuration): k_throw is not launched at
    // all -- an empty launch still costs ~8 us of the exposure's critical path -- and the lanes take what they
    // find up to kLaneReach electrons; a bin beyond that (the estimate carries no Poisson noise) sets status bit 1
    // and the exposure is run again with k_throw when its status is read (check_status)
    // (knob `lane_reach`: a test knob that lowers the lanes' reach so that the re-run path can be exercised)
    int reach = kLaneReach;
    if (c->knobs.lane_reach >= 0) reach = (int)std::min<long long>(std::max<long long>(c->knobs.lane_reach, 1), kLaneReach);
    lane_unlimited = d.rng_mode == WAYNE_RNG_SPLIT && s.est_thrown <= 0. && d.thrower_splits <= 0 && !s.force_throw &&
                     c->knobs.throw_wgs < 0;
    a.lane_max = lane_unlimited ? reach : kLaneMax;
    a.prefix = s.prefix.as<uint32_t>(); a.xpos = s.xpos.as<double>(); a.ypos = s.ypos.as<double>();
    a.sub = s.sub.as<SubInfo>();
    a.total_electrons = c->counters.as<unsigned long long>();
    a.status = (int*)(s.misc.as<char>() + 8);
    const int n_chunks = (W + kPrepThreads - 1) / kPrepThreads;
    a.chunk_total = s.chunk_total.as<uint32_t>();
    a.chunk_box = s.chunk_box.as<double>();
    a.fix_inline = lane_unlimited ? 1 : 0;
    // a finely sampled scan holds a few electrons per bin and sub-sample: no bin is expected to reach the
    // multinomial's threshold (mean <= 6 against kSplitMin = 32: 1e-13 per draw), k_narrow's launch would only find
    // that out workgroup by workgroup (0.04 ms at K = 2233) -- it is left out, and a bin that qualifies after all
    // flags the run, which is then repeated with every kernel (check_status)
    skip_narrow = lane_unlimited && s.max_narrow <= 6. && !(c->knobs.keep_narrow > 0);
    a.no_narrow = skip_narrow ? 1 : 0;
    CosmicArgs& ca = cosmic_args;
    ca.R = R; ca.N = N; ca.S = S; ca.seed = d.seed; ca.exposure = d.exposure_index;
    ca.rate = (d.cosmic_rate >= 0.) ? d.cosmic_rate : -1.;
    ca.read_dt = s.read_dt.as<double>(); ca.acc = s.acc.as<long long>();
    ca.seg = s.seg.as<uint32_t>();
    // thin exposure, nothing for k_throw or k_narrow expected: the lanes plan their bins themselves (k_lane, FUSED)
    fused = lane_unlimited && skip_narrow && s.thin && !(c->knobs.no_fuse > 0);
    s.fused_last = fused;
    s.last_prep = a;
    s.last_chunks = n_chunks;
    if (!fused) {
      ProfScope ps(c, PK_PREP_SUB);
      hipLaunchKernelGGL(k_prep_sub, dim3(K, n_chunks), dim3(kPrepThreads), 0, c->stream, a, ca);
      HIP_TRY(c, hipGetLastError());
      if (!a.fix_inline) {
        hipLaunchKernelGGL(k_prep_fix, dim3(K), dim3(kPrepThreads), 0, c->stream, a, n_chunks);
        HIP_TRY(c, hipGetLastError());
      }
    }
  }
  {
    ThrowArgs a{};
    a.W = W; a.K = K; a.N = N; a.S = S;
    a.kb = s.kb;
    // grid: one unit (128 electrons; 1 in replay mode) per lane of the workgroups of a sub-sample, from
    // the host's estimate of the electrons + 8 %, but at least ~4 workgroups per CU over the launch
    // (knob `throw_wgs` / desc.thrower_splits override); k_throw shares out what it actually finds
    const int min_wgs = 1024;
    int splits = d.thrower_splits;
    if (splits <= 0) {
      if (c->knobs.throw_wgs >= 0) {
        splits = (int)std::max<long long>(1, (std::max<long long>(c->knobs.throw_wgs, 1) + K - 1) / K);
      } else {
        const double unit = (d.rng_mode == WAYNE_RNG_REPLAY) ? 1. : (double)kThrowBlock;
        const double lanes = 1.08 * s.est_thrown / unit;
        splits = (int)std::min(4096., std::ceil(lanes / kThrowThreads));
        // split mode with no bin expected beyond a lane's cap: k_throw finds nothing to do (any stray bin is
        // handled by the one workgroup per sub-sample launched here)
        if (d.rng_mode == WAYNE_RNG_SPLIT && s.est_thrown <= 0. && !s.force_throw) splits = -1;
        // a lane takes several units when the launch would exceed ~24 workgroups per CU: then ~12 per CU
        // (each lane a handful of units) is the measured optimum (scripts/sweep_throw.py)
        const int cap = std::max(1, (3072 + K - 1) / K);
        if (splits > 2 * cap) splits = cap;
        splits = splits < 0 ? 1 : std::max(splits, (min_wgs + K - 1) / K);
      }
    }
    a.min_wgs = min_wgs;
    a.splits = std::min(splits, 4096);
    a.margin = margin;
    const int lds_ints = thrower_lds_ints(c, a.splits, margin);
    a.lds_ints = lds_ints;
    a.threads_compat = d.threads_compat;
    a.seed = d.seed; a.exposure = d.exposure_index; a.subsample0 = 0;
    a.flags = d.flags;
    // grism.py:363: indices + (1014 - size) / 2 -- the same number as the frame offset 507 - SUBARRAY / 2
    // (exposure_generator.py:630) for every sub-array, so the descriptor's sub_scale serves both: 0 for the
    // full array, or the reference's -5 there when the caller keeps its quirks (the host then uploads the
    // flat planes rolled by +5 px, numpy's wrap-around for the negative indices)
    a.flat_off = d.sub_scale;
    a.flat_wmin = c->g.flat_wmin; a.flat_wmax = c->g.flat_wmax;
    a.flat_inv_range = 1.0 / (c->g.flat_wmax - c->g.flat_wmin);
    a.sub = s.sub.as<SubInfo>(); a.prefix = s.prefix.as<uint32_t>(); a.nwide = s.nwide.as<int32_t>();
    a.nsplit = s.nsplit.as<int32_t>();
    a.nlane = s.nlane.as<int32_t>();
    std::memcpy(a.chunk_order, s.chunk_order, sizeof a.chunk_order);
    std::memcpy(a.lane_order, s.lane_order, sizeof a.lane_order);
    a.xpos = s.xpos.as<double>(); a.ypos = s.ypos.as<double>();
    a.sigl = s.sigl.as<double>(); a.sigh = s.sigh.as<double>();
    for (int i = 0; i < 4; ++i) a.flat[i] = c->has_flat ? c->flat[i].as<float>() : nullptr;
    a.acc = s.acc.as<long long>();
    a.frame = nullptr;
    if ((d.flags & WAYNE_F_ADD_FLAT) && !c->has_flat) return fail(c, WAYNE_E_STATE, "run: add_flat without a flat cube");
    const int si_ = slot % c->n_streams;
    const bool fork = d.rng_mode == WAYNE_RNG_SPLIT && c->fork_narrow && !skip_narrow;
    hipStream_t main_stream = c->stream;
    {
      // (with the fork, the PK_THROW interval spans all thrower kernels; PK_NARROW / PK_LANE are those kernels' own)
      ProfScope ps_throw(c, PK_THROW);
      if (fork) {
        HIP_TRY(c, hipEventRecord(c->ev_fork[si_], main_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->side[si_], c->ev_fork[si_], 0));
        c->stream = c->side[si_];
        int rc;
        {
          ProfScope ps(c, PK_NARROW);
          rc = launch_narrow<1>(c, a, (d.flags & WAYNE_F_EXACT_SAMPLERS) != 0);
        }
        if (rc == WAYNE_OK && hipEventRecord(c->ev_join[si_], c->side[si_]) != hipSuccess)
          rc = fail(c, WAYNE_E_HIP, "run: event record on the side stream");
        c->stream = main_stream;
        if (rc) return rc;
      }
      int rc = lane_unlimited ? (int)WAYNE_OK
               : (d.rng_mode == WAYNE_RNG_REPLAY) ? launch_throw<0, 1>(c, a, lds_ints) : launch_throw<1, 1>(c, a, lds_ints);
      if (rc) return rc;
      if (d.rng_mode == WAYNE_RNG_SPLIT) {
        ProfScope ps(c, PK_LANE);
        if ((rc = fused ? launch_lane<1>(c, a, true, &prep_args, &cosmic_args) : launch_lane<1>(c, a, s.thin))) return rc;
      }
      if (fork) HIP_TRY(c, hipStreamWaitEvent(main_stream, c->ev_join[si_], 0));
    }
    if (!fork && d.rng_mode == WAYNE_RNG_SPLIT && !skip_narrow) {
      ProfScope ps(c, PK_NARROW);
      int rc = launch_narrow<1>(c, a, (d.flags & WAYNE_F_EXACT_SAMPLERS) != 0);
      if (rc) return rc;
    }
  }
  s.acc_dirty = true;
  s.front_done = true;
  return WAYNE_OK;
}

int wayne_exposure_run_back(wayne_ctx* c, int slot) {
  if (!c) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "run: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded || !s.front_done) return fail(c, WAYNE_E_STATE, "run_back: run_front first");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  const wayne_exposure_desc& d = s.d;
  const int S = c->S;
  RampArgs a{};
  a.R = s.R; a.N = c->N; a.S = S;
#ifdef WAYNE_TIMING_KNOBS
  // measurement knob of TIMING BUILDS only (scripts/ramp_vs_reads.py builds its own library with -DWAYNE_TIMING_KNOBS):
  // the kernel works through the first n reads only -- the slot's other accumulators stay uncleared and the later
  // reads' planes stale, so the shipped library does not look at the knob at all
  bool ramp_reads_cut = false;
  if (c->knobs.ramp_reads >= 0) { a.R = (int)std::max<long long>(1, std::min<long long>(s.R, c->knobs.ramp_reads)); ramp_reads_cut = a.R < s.R; }
#endif
  a.seed = d.seed; a.exposure = d.exposure_index; a.flags = d.flags;
  a.sky_ct_s = d.sky_ct_s; a.noise_mean = d.noise_mean; a.noise_std = d.noise_std;
  a.read_dt = s.read_dt.as<double>();
  a.acc = s.acc.as<long long>();
  a.pfl = c->has_pfl ? c->pfl.as<float>() : nullptr;
  a.sky = c->has_sky ? c->sky.as<float>() : nullptr;
  for (int i = 0; i < 4; ++i) a.lin[i] = c->has_lin ? c->lin[i].as<float>() : nullptr;
  a.dark_sci = c->has_dark ? c->dark_sci.as<float>() : nullptr;
  a.dark_err = c->has_dark ? c->dark_err.as<float>() : nullptr;
  a.zero_read = c->has_zero ? c->zero_read.as<double>() : nullptr;
  a.out = s.out.p;
  if ((d.flags & WAYNE_F_ADD_GAIN_VARIATIONS) && !c->has_pfl) return fail(c, WAYNE_E_STATE, "run: add_gain_variations without a pixel flat");
  if ((d.flags & WAYNE_F_ADD_NON_LINEAR) && !c->has_lin) return fail(c, WAYNE_E_STATE, "run: add_non_linear without coefficient planes");
  if (d.sky_ct_s > 0. && !c->has_sky) return fail(c, WAYNE_E_STATE, "run: sky background without a master sky");
  // sky tables were planned and uploaded with the descriptor (prepare_sky_tables)
  a.sky_alias = s.sky_alias_on ? s.sky_tab.as<uint32_t>() : nullptr;
  a.alias_mask = s.sky_alias_on ? s.sky_mask : 0u;
  a.sky_levels = s.sky_L;
  for (int l = 0; l < 16; ++l) { a.sky_level[l] = s.sky_level[l]; a.sky_tab0[l] = s.sky_tab0[l]; a.tab0[l] = s.sky_tab0[l]; a.bg[l] = 0.f; }
  for (int r = 0; r < s.R && r < 16; ++r) a.bg[r] = (float)(d.sky_ct_s * s.read_dt_host[r]);
  a.use_box = s.use_box ? 1 : 0;
  std::memcpy(a.box, s.acc_box, sizeof a.box);
  a.seg = s.seg.as<uint32_t>();
  const int threads = kRampThreads;
  const unsigned blocks = (unsigned)(((size_t)S * S + threads - 1) / threads);
  {
    ProfScope ps(c, PK_RAMP, true);
    void (*kern)(RampArgs) = select_ramp(c, s, nullptr);
    if (ps.on) hipExtLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, c->stream, ps.rec.a, ps.rec.b, 0, a);
    else hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
  }
  s.acc_dirty = false;
#ifdef WAYNE_TIMING_KNOBS
  if (ramp_reads_cut) s.acc_dirty = true;     // the next front half starts from cleared accumulators
#endif
  s.front_done = false;
  s.ran = true;
  return WAYNE_OK;
}

int wayne_exposure_ramp_variant(wayne_ctx* c, int slot, char* buf, int cap) {
  if (!c || !buf || cap <= 0) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "ramp_variant: slot");
  const Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "ramp_variant: slot not uploaded");
  std::string name;
  (void)select_ramp(c, s, &name);
  std::snprintf(buf, (size_t)cap, "%s", name.c_str());
  return WAYNE_OK;
}

int wayne_exposure_run(wayne_ctx* c, int slot) {
  int rc = wayne_exposure_run_front(c, slot);
  if (rc) return rc;
  return wayne_exposure_run_back(c, slot);
}

int wayne_exposure_status(wayne_ctx* c, int slot, int* status) {
  if (!c || !status) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "status: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "status: slot not uploaded");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  Slot::Misc m{};
  HIP_TRY(c, hipMemcpyAsync(&m, s.misc.p, sizeof m, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *status = m.status;
  return WAYNE_OK;
}

unsigned long long wayne_ctx_reruns(const wayne_ctx* c) { return c ? (unsigned long long)c->reruns : 0ull; }

// Status word of the slot's last run: bit 0 = overflow (an error), bit 1 = a bin beyond the lanes' reach in an
// exposure launched without k_throw (*rerun is set: the caller runs the exposure again, now with k_throw).
static int check_status(wayne_ctx* c, Slot& s, bool* rerun = nullptr) {
  struct { unsigned long long electrons; int status; int pad; } m{};
  HIP_TRY(c, hipMemcpyAsync(&m, s.misc.p, sizeof m, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  s.ran = false;                           // looked at
  if (m.status & 1) return fail(c, WAYNE_E_OVERFLOW, "exposure: a sub-sample holds >= 2^32 electrons (or a bin >= 2^31)");
  if (rerun) *rerun = (m.status & 2) != 0;
  return WAYNE_OK;
}

int wayne_exposure_run_checked(wayne_ctx* c, int slot) {
  int rc = wayne_exposure_run(c, slot);
  if (rc) return rc;
  Slot& s = c->slots[slot];
  bool rerun = false;
  if ((rc = check_status(c, s, &rerun)) || !rerun) return rc;
  s.force_throw = true;
  c->reruns += 1;
  if ((rc = wayne_exposure_run(c, slot))) return rc;
  return check_status(c, s);
}

int wayne_exposure_download(wayne_ctx* c, int slot, void* out_reads) {
  if (!c || !out_reads) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "download: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "download: slot not uploaded");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  const size_t SS = (size_t)c->S * c->S;
  const size_t out_elem = (s.d.flags & WAYNE_F_OUT_F64) ? sizeof(double) : sizeof(float);
  HIP_TRY(c, hipMemcpyAsync(out_reads, s.out.p, (size_t)(s.R + 1) * SS * out_elem, hipMemcpyDeviceToHost, c->stream));
  bool rerun = false;
  int rc = check_status(c, s, &rerun);
  if (rc || !rerun) return rc;
  s.force_throw = true;
  c->reruns += 1;
  if ((rc = wayne_exposure_run(c, slot))) return rc;
  HIP_TRY(c, hipMemcpyAsync(out_reads, s.out.p, (size_t)(s.R + 1) * SS * out_elem, hipMemcpyDeviceToHost, c->stream));
  return check_status(c, s);
}

int wayne_exposure_fetch_async(wayne_ctx* c, int slot) {
  if (!c) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "fetch_async: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "fetch_async: slot not uploaded");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  const size_t SS = (size_t)c->S * c->S;
  const size_t bytes = (size_t)(s.R + 1) * SS * ((s.d.flags & WAYNE_F_OUT_F64) ? sizeof(double) : sizeof(float));
  const size_t tail = (bytes + 63) & ~(size_t)63;
  if (s.pinned_cap < tail + 64) {
    if (s.pinned) (void)hipHostFree(s.pinned);
    s.pinned = nullptr;
    s.pinned_cap = 0;
    if (hipHostMalloc(&s.pinned, tail + 64, hipHostMallocDefault) != hipSuccess)
      return fail(c, WAYNE_E_NOMEM, "fetch_async: pinned host allocation failed");
    s.pinned_cap = tail + 64;
  }
  s.pinned_misc = (Slot::Misc*)((char*)s.pinned + tail);
  // The copy follows the slot's kernels on the slot's own stream: while it runs (1.2 ms at 55 GB/s for a full
  // frame) the kernels of the exposure in the next slot run on the other stream.  (A separate copy stream fed by
  // events was measured: 660-700 exposures/s instead of 810-826 -- scripts/probe_pipeline.py.)
  if (c->n_streams == 2 && c->ev_kdone[slot % 2]) {
    HIP_TRY(c, hipEventRecord(c->ev_kdone[slot % 2], c->stream));
    c->kdone_valid[slot % 2] = true;
  }
  HIP_TRY(c, hipMemcpyAsync(s.pinned, s.out.p, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(s.pinned_misc, s.misc.p, sizeof(Slot::Misc), hipMemcpyDeviceToHost, c->stream));
  return WAYNE_OK;
}

int wayne_exposure_wait(wayne_ctx* c, int slot, void** host_reads) {
  if (!c || !host_reads) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "wait: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded || !s.pinned || !s.pinned_misc) return fail(c, WAYNE_E_STATE, "wait: fetch_async first");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *host_reads = s.pinned;
  s.ran = false;                             // looked at (the status word came with the reads)
  if (s.pinned_misc->status & 1)
    return fail(c, WAYNE_E_OVERFLOW, "exposure: a sub-sample holds >= 2^32 electrons (or a bin >= 2^31)");
  if (s.pinned_misc->status & 2) {           // a bin beyond the lanes' reach: once more, with k_throw
    s.force_throw = true;
    c->reruns += 1;
    int rc = wayne_exposure_run(c, slot);
    if (rc == WAYNE_OK) rc = wayne_exposure_fetch_async(c, slot);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    s.ran = false;
    if (s.pinned_misc->status & 1)
      return fail(c, WAYNE_E_OVERFLOW, "exposure: a sub-sample holds >= 2^32 electrons (or a bin >= 2^31)");
  }
  return WAYNE_OK;
}

void* wayne_exposure_device_reads(wayne_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= kSlots) return nullptr;
  return c->slots[slot].out.p;
}

int wayne_exposure_synthesize(wayne_ctx* c, const wayne_exposure_desc* d, void* out_reads) {
  int rc = wayne_exposure_upload(c, 0, d);
  if (rc) return rc;
  if ((rc = wayne_exposure_run(c, 0))) return rc;
  return wayne_exposure_download(c, 0, out_reads);
}

int wayne_exposure_debug_fetch(wayne_ctx* c, int slot, int32_t* counts, double* x_pos, double* y_pos,
                               double* acc_e) {
  if (!c) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "debug_fetch: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "debug_fetch: slot not uploaded");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  const size_t KW = (size_t)s.K * s.W;
  if (counts) HIP_TRY(c, hipMemcpyAsync(counts, s.counts.p, KW * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (s.fused_last && (x_pos || y_pos)) {
    // a fused front half kept no positions: k_prep_sub works them out now (same code, same inputs; its cosmic-ray
    // stage off and its electron count into a spare counter)
    PrepArgs a = s.last_prep;
    a.total_electrons = c->counters.as<unsigned long long>() + 1;
    CosmicArgs off{};
    off.rate = -1.;
    hipLaunchKernelGGL(k_prep_sub, dim3(s.K, s.last_chunks), dim3(kPrepThreads), 0, c->stream, a, off);
    HIP_TRY(c, hipGetLastError());
  }
  if (x_pos) HIP_TRY(c, hipMemcpyAsync(x_pos, s.xpos.p, KW * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (y_pos) HIP_TRY(c, hipMemcpyAsync(y_pos, s.ypos.p, KW * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  {
    // (between run_front and run_back: a run that met a bin beyond the lanes' reach is repeated with k_throw first)
    bool rerun = false;
    int rc = check_status(c, s, &rerun);
    if (rc) return rc;
    if (rerun && s.front_done) {
      s.force_throw = true;
      c->reruns += 1;
      if ((rc = wayne_exposure_run_front(c, slot))) return rc;
      return wayne_exposure_debug_fetch(c, slot, counts, x_pos, y_pos, acc_e);
    }
  }
  if (acc_e) {
    const size_t n = (size_t)s.R * c->S * c->S;
    std::vector<long long> tmp(n);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), s.acc.p, n * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) acc_e[i] = (double)tmp[i] * kInvQ;
  }
  return check_status(c, s);
}

int wayne_exposure_debug_boxes(wayne_ctx* c, int slot, int32_t* boxes, int32_t* segments, int* use_box) {
  if (!c || !boxes || !segments || !use_box) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "debug_boxes: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded) return fail(c, WAYNE_E_STATE, "debug_boxes: slot not uploaded");
  *use_box = s.use_box ? 1 : 0;
  const int S = c->S;
  for (int r = 0; r < 16; ++r) {
    for (int i = 0; i < 4; ++i) boxes[4 * r + i] = s.use_box ? s.acc_box[r][i] : 0;
    int n = 0;
    if (r < s.R) {
      // the wave test of k_ramp (acc_live) over the frame's 64-pixel segments
      for (int p0 = 0; p0 < S * S; p0 += 64) {
        const int wy0 = p0 / S, wy1 = std::min(p0 + 63, S * S - 1) / S, wx0 = p0 - wy0 * S;
        const int* b = s.acc_box[r];
        const bool rows = wy0 < b[3] && wy1 >= b[2];
        const bool cols = (wy0 != wy1) || (wx0 < b[1] && wx0 + 64 > b[0]);
        n += (!s.use_box || (rows && cols)) ? 1 : 0;
      }
    }
    segments[r] = n;
  }
  return WAYNE_OK;
}

int wayne_exposure_debug_depth(wayne_ctx* c, int slot, double* depth) {
  if (!c || !depth) return WAYNE_E_INVALID;
  if (slot < 0 || slot >= kSlots) return fail(c, WAYNE_E_INVALID, "debug_depth: slot");
  Slot& s = c->slots[slot];
  if (!s.uploaded || !s.has_depth) return fail(c, WAYNE_E_STATE, "debug_depth: no depth matrix in this slot");
  (void)hipSetDevice(c->device);
  use_slot_stream(c, slot);
  HIP_TRY(c, hipMemcpyAsync(depth, s.depth.p, (size_t)s.K * s.W * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return WAYNE_OK;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
int wayne_profile_enable(wayne_ctx* c, int on) {
  if (!c) return WAYNE_E_INVALID;
  c->prof_on = on != 0;
  return WAYNE_OK;
}

int wayne_profile_select(wayne_ctx* c, unsigned mask) {
  if (!c) return WAYNE_E_INVALID;
  c->prof_mask = mask;
  return WAYNE_OK;
}

int wayne_profile_reset(wayne_ctx* c) {
  if (!c) return WAYNE_E_INVALID;
  int rc = collect_profile(c);
  if (rc) return rc;
  for (int i = 0; i < WAYNE_PROF_KERNELS; ++i) { c->prof_launches[i] = 0; c->prof_ms[i] = 0; }
  c->electrons = 0;
  HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, kCounterBytes, c->streams[0]));
  return sync_all(c);
}

int wayne_profile_get(wayne_ctx* c, wayne_profile* out) {
  if (!c || !out) return WAYNE_E_INVALID;
  int rc = collect_profile(c);
  if (rc) return rc;
  for (int i = 0; i < WAYNE_PROF_KERNELS; ++i) {
    out->name[i] = kProfNames[i];
    out->launches[i] = c->prof_launches[i];
    out->ms[i] = c->prof_ms[i];
  }
  unsigned long long words[kCounterStripes * kCounterStride], dev = 0;
  HIP_TRY(c, hipMemcpy(words, c->counters.p, sizeof words, hipMemcpyDeviceToHost));
  for (int i = 0; i < kCounterStripes; ++i) dev += words[i * kCounterStride];
  out->electrons = c->electrons + dev;
  return WAYNE_OK;
}

void wayne_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}

void wayne_host_sample_draws(uint32_t seed, uint32_t exposure, int n_samples, double* z_x, double* z_y,
                             int32_t* rand_seed) {
  for (int k = 0; k < n_samples; ++k) {
    const u32x4 w = philox4x32_10((uint32_t)k, 0u, 0u, exposure, seed, STAGE_HOST);
    const double ua = u01d(w.v[0]), ub = u01d(w.v[1]);
    const double R = std::sqrt(-2.0 * std::log(ub));
    const double ang = 6.283185307179586476925 * ua;
    if (z_x) z_x[k] = R * std::cos(ang);
    if (z_y) z_y[k] = R * std::sin(ang);
    if (rand_seed) rand_seed[k] = (int32_t)uint_below(w.v[2], 100000u);
  }
}

}  // extern "C"
