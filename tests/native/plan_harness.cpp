// CPU harness of the product's host-side launch planner (wayne_amd/csrc/host_plan.h -- the SAME file wayne_hip.hip
// includes), built with g++ -fsanitize=address,undefined -fno-sanitize-recover=all (tests/native/Makefile) and driven by
// tests/test_host_plan.py through a batch file: a sequence of operations in, a sequence of results out.  Test
// infrastructure: nothing in wayne_amd/ loads or runs it.
//
//   plan_harness <in> <out>
//
// Operations (little-endian; i32 code first):
//   1 SET_GRISM  f64 trace[9] wlsol[9] p_ratio[4] p_sigl[4] p_sigh[4]; i32 n_sens; f64 sens_wl[n] sens_val[n]   -> i32 table_ok
//   2 PLAN       i32 S sub_scale rng_mode W K R; f64 scale_factor; f64 wl[W] flux[W] x_ref[K] y_ref[K] dur_ms[K]; i32 sample_read[K]
//   3 SKY        f64 sky_ct_s; i32 R; f64 read_dt[R]; i32 has_sky n_sorted; f32 sky_sorted[n]
//   4 ALIAS      f64 lam
//   5 PSF        i32 size N rng_mode threads_compat margin; i32 counts[size]; f64 x[size] y[size] ratio[size] sigl[size]
// Results: see the write_* calls below (tests/plan_harness.py decodes them).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../wayne_amd/csrc/host_plan.h"

using namespace wayne;

namespace {
FILE* fin = nullptr;
FILE* fout = nullptr;

template <class T> bool get(T* v, size_t n = 1) { return n == 0 || std::fread(v, sizeof(T), n, fin) == n; }
template <class T> void put(const T* v, size_t n = 1) { if (n && std::fwrite(v, sizeof(T), n, fout) != n) { std::perror("write"); std::exit(3); } }
template <class T> void put1(T v) { put(&v, 1); }
[[noreturn]] void bad(const char* what) { std::fprintf(stderr, "plan_harness: malformed input (%s)\n", what); std::exit(2); }
}  // namespace

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: plan_harness <in> <out>\n"); return 2; }
  fin = std::fopen(argv[1], "rb");
  fout = std::fopen(argv[2], "wb");
  if (!fin || !fout) { std::perror("open"); return 2; }
  plan::SpectrumEstimate est;
  int32_t code;
  long ops = 0;
  while (get(&code)) {
    ++ops;
    if (code == 1) {
      GrismDev g{};
      int32_t n = 0;
      if (!get(g.trace, 9) || !get(g.wlsol, 9) || !get(g.p_ratio, 4) || !get(g.p_sigl, 4) || !get(g.p_sigh, 4) || !get(&n))
        bad("grism");
      if (n < 0 || n > (1 << 24)) bad("n_sens");
      std::vector<double> swl((size_t)n), sval((size_t)n);
      if (!get(swl.data(), (size_t)n) || !get(sval.data(), (size_t)n)) bad("sensitivity");
      g.n_sens = n;
      // what wayne_ctx_set_grism answers (a table that fails is refused there) -- and the planner takes the table either
      // way: whatever it holds, the interpolation must stay inside it
      const bool ok = plan::sens_table_ok(swl.data(), sval.data(), n);
      est.set_grism(g, swl.data(), sval.data(), n);
      put1<int32_t>(1);
      put1<int32_t>(ok ? 1 : 0);
    } else if (code == 2) {
      int32_t h[6];
      double scale = 1.;
      if (!get(h, 6) || !get(&scale)) bad("plan header");
      const int S = h[0], sub_scale = h[1], rng_mode = h[2], W = h[3], K = h[4], R = h[5];
      if (W < 0 || K < 0 || W > (1 << 22) || K > (1 << 22)) bad("plan sizes");
      // exact-size heap blocks: a read one element beyond any of them is an AddressSanitizer report
      std::vector<double> wl((size_t)W), flux((size_t)W), xr((size_t)K), yr((size_t)K), dur((size_t)K);
      std::vector<int32_t> sread((size_t)K);
      if (!get(wl.data(), (size_t)W) || !get(flux.data(), (size_t)W) || !get(xr.data(), (size_t)K) || !get(yr.data(), (size_t)K) ||
          !get(dur.data(), (size_t)K) || !get(sread.data(), (size_t)K)) bad("plan arrays");
      plan::ThrowPlan tp;
      plan::estimate_thrown(est, W, wl.data(), flux.data(), K, dur.data(), scale, rng_mode, &tp);
      int kb = 1;
      bool thin = false;
      plan::lane_batches(K, W, tp.max_chunk_electrons, &kb, &thin);
      int box[16][4];
      for (auto& b : box) b[0] = b[1] = b[2] = b[3] = -12345;
      const bool use = plan::accumulator_boxes(est, W, wl.data(), flux.data(), K, R, S, sub_scale, xr.data(), yr.data(),
                                               sread.data(), box);
      put1<int32_t>(2);
      put1<int32_t>(use ? 1 : 0);
      for (auto& b : box) { int32_t v[4] = {b[0], b[1], b[2], b[3]}; put(v, 4); }
      put1<double>(tp.est_thrown); put1<double>(tp.max_chunk_electrons); put1<double>(tp.max_narrow);
      put1<int32_t>(tp.n_chunks); put1<int32_t>(tp.n_lane_chunks);
      put(tp.chunk_order, (size_t)kMaxChunks); put(tp.lane_order, (size_t)kMaxChunks);
      put1<int32_t>(kb); put1<int32_t>(thin ? 1 : 0);
      put1<double>(est.smax); put1<double>(est.wl_lo); put1<double>(est.wl_hi);
      put1<int32_t>(est.sig_ok ? 1 : 0);
      put1<int64_t>((int64_t)est.rebuilds);
      // the per-bin factors too (cache-key tests compare them with a fresh planner's)
      put(est.rate.data(), est.rate.size()); put(est.ratio.data(), est.ratio.size()); put(est.sigl.data(), est.sigl.size());
    } else if (code == 3) {
      double sky_ct_s = 0.;
      int32_t R = 0, has_sky = 0, n = 0;
      if (!get(&sky_ct_s) || !get(&R)) bad("sky header");
      if (R < 0 || R > 64) bad("sky R");
      std::vector<double> dt((size_t)R);
      if (!get(dt.data(), (size_t)R) || !get(&has_sky) || !get(&n)) bad("sky reads");
      if (n < 0 || n > (1 << 24)) bad("sky n");
      std::vector<float> sorted((size_t)n);
      if (!get(sorted.data(), (size_t)n)) bad("sky pixels");
      const float smin = n ? sorted.front() : 0.f, smax = n ? sorted.back() : 0.f;
      plan::SkyPlan sp;
      plan::plan_sky(sky_ct_s, R, dt.data(), has_sky != 0, smin, smax, sorted, &sp);
      put1<int32_t>(3);
      put1<int32_t>(sp.alias_on ? 1 : 0); put1<int32_t>(sp.pieces ? 1 : 0);
      put1<uint32_t>(sp.mask); put1<int32_t>(sp.L);
      put(sp.level, 16); put(sp.tab0, 16);
      put1<int32_t>((int32_t)sp.keys.size());
      put(sp.keys.data(), sp.keys.size());
      put1<int32_t>(sp.n_bg);
      if (sp.alias_on)
        for (uint32_t key : sp.keys) {
          float lam;
          std::memcpy(&lam, &key, 4);
          uint32_t tab[kSkyAlias];
          plan::build_sky_alias((double)lam, tab);
          put(tab, (size_t)kSkyAlias);
        }
    } else if (code == 4) {
      double lam = 0.;
      if (!get(&lam)) bad("alias");
      uint32_t tab[kSkyAlias];
      plan::build_sky_alias(lam, tab);
      put1<int32_t>(4);
      put1<int32_t>(plan::sky_alias_fits(lam) ? 1 : 0);
      put(tab, (size_t)kSkyAlias);
    } else if (code == 5) {
      int32_t h[5];
      if (!get(h, 5)) bad("psf header");
      const int size = h[0];
      if (size < 0 || size > (1 << 22)) bad("psf size");
      std::vector<int32_t> counts((size_t)size);
      std::vector<double> x((size_t)size), y((size_t)size), ratio((size_t)size), sigl((size_t)size);
      if (!get(counts.data(), (size_t)size) || !get(x.data(), (size_t)size) || !get(y.data(), (size_t)size) ||
          !get(ratio.data(), (size_t)size) || !get(sigl.data(), (size_t)size)) bad("psf arrays");
      plan::PsfPlan pp;
      const int rc = plan::plan_psf_apply(counts.data(), size, x.data(), y.data(), ratio.data(), sigl.data(), h[1], h[2], h[3],
                                          h[4], &pp);
      put1<int32_t>(5);
      put1<int32_t>(rc);
      put1<int64_t>((int64_t)pp.total);
      if (rc == 0) {
        put(pp.prefix.data(), pp.prefix.size());
        put(pp.nwide.data(), pp.nwide.size()); put(pp.nsplit.data(), pp.nsplit.size()); put(pp.nlane.data(), pp.nlane.size());
        put1<int32_t>(pp.any_split ? 1 : 0); put1<int32_t>(pp.any_lane ? 1 : 0);
        put1<uint32_t>(pp.run);
        int32_t rect[4] = {pp.tx0, pp.ty0, pp.tw, pp.th};
        put(rect, 4);
      }
    } else {
      bad("operation code");
    }
  }
  std::fclose(fin);
  if (std::fclose(fout) != 0) { std::perror("close"); return 3; }
  std::fprintf(stderr, "plan_harness: %ld operations\n", ops);
  return 0;
}
