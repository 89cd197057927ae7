/*
 * wayne_hip.h -- C ABI of libwayne_hip.so, the MI355X (gfx950) WFC3-IR
 * exposure-synthesis path.  Plain pointers and sizes only; bound from Python
 * with ctypes (wayne_amd/_lib.py) and from anything else that can call C.
 *
 * Each entry point names the reference interface it replaces
 * (file:line under the ucl-exoplanets/wayne tree).
 *
 * Conventions
 *   - every function returning int returns WAYNE_OK (0) or a negative
 *     WAYNE_E_* code; wayne_last_error(ctx) gives the message;
 *   - host pointers are borrowed for the duration of the call only;
 *   - the library owns all device memory; one wayne_ctx per GPU / stream;
 *     calls on distinct contexts are thread-safe, calls on one are not;
 *   - frames are row-major, y-major: pixel (y, x) at [y * side + x], exactly
 *     as the reference's `pixel_array[ypos*nc + xpos]` (pyparallel_menu.c:94).
 *   - N = light-sensitive side (SUBARRAY, or 1014 for SUBARRAY 1024),
 *     S = N + 10 = side with the 5-px reference border (detector.py:102-124).
 */
#ifndef WAYNE_HIP_H
#define WAYNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAYNE_ABI_VERSION 7

/* status codes */
#define WAYNE_OK 0
#define WAYNE_E_INVALID (-1)   /* bad argument / shape                       */
#define WAYNE_E_NEGATIVE (-2)  /* negative electron count                     */
#define WAYNE_E_OVERFLOW (-3)  /* sum(counts)*threads >= 2^31 in replay mode   */
#define WAYNE_E_NOMEM (-4)     /* host or device allocation failed             */
#define WAYNE_E_HIP (-5)       /* a HIP runtime call failed                    */
#define WAYNE_E_NODEVICE (-6)  /* no usable gfx950 device                      */
#define WAYNE_E_STATE (-7)     /* grism / calibration / upload missing         */

/* rng_mode */
#define WAYNE_RNG_REPLAY 0 /* glibc rand_r streams + OpenMP partition of the reference: bit-exact */
#define WAYNE_RNG_PHILOX 1 /* Philox-keyed streams, every electron thrown individually */
#define WAYNE_RNG_SPLIT 2  /* production default: a bin with >= 32 narrow electrons has its narrow
                              component drawn as ONE multinomial (binomial chains, k_narrow); what is left
                              to throw one by one -- its wide electrons, or the whole of a bin that does
                              not qualify -- is thrown by the bin's own lane from the bin's own stream
                              (k_lane), or, beyond 4096 such electrons, shared out as in mode 1.
                              Same distribution of the frame, several times less work            */

/* wayne_exposure_desc.flags -- the keyword switches of
 * ExposureGenerator.scanning_frame (exposure_generator.py:178-192) */
#define WAYNE_F_ADD_FLAT (1u << 0)
#define WAYNE_F_ADD_GAIN_VARIATIONS (1u << 1)
#define WAYNE_F_ADD_NON_LINEAR (1u << 2)
#define WAYNE_F_CLIP_DET_LIMITS (1u << 3)
#define WAYNE_F_ADD_READ_NOISE (1u << 4)
#define WAYNE_F_ADD_STELLAR_NOISE (1u << 5)
#define WAYNE_F_ADD_DARK (1u << 6)
#define WAYNE_F_ADD_INITIAL_BIAS (1u << 7)
#define WAYNE_F_OUT_F64 (1u << 16) /* reads delivered as float64 (the reference's dtype) instead of float32 */
#define WAYNE_F_EXACT_SAMPLERS (1u << 17) /* IEEE divide/sqrt + libm-grade log/exp/sin/cos in the per-pixel
                                             Poisson / normal draws (parity runs) instead of the hardware
                                             approximations (production); same algorithm, same streams */

typedef struct wayne_ctx wayne_ctx;

/* ---- context ---------------------------------------------------------- */

int wayne_abi_version(void);
const char *wayne_strerror(int status);
/* The compile-time switches (-DWAYNE_...) of this build that change what the library computes or launches, separated
 * by spaces: "" for the shipped library.  Negative-control and timing builds (tests/native, scripts/) name theirs
 * here, so that a measurement or a file can say which library produced it and refuse a defective one. */
const char *wayne_build_flags(void);

/* Number of HIP devices visible (0 when there is none / no driver). */
int wayne_device_count(void);

/* Create a context on `device` with its own HIP stream.  NULL on failure
 * (*status, if given, says why).  Fails -- never falls back to the CPU --
 * when no gfx950 GPU is present. */
wayne_ctx *wayne_ctx_create(int device, int *status);
void wayne_ctx_destroy(wayne_ctx *ctx);
const char *wayne_last_error(const wayne_ctx *ctx);
/* Wait for everything enqueued on the context AND settle it: the status word of every exposure run since the last
 * look is read (one copy for all slots); an exposure that met a bin beyond the reach of the short launch sequence
 * chosen from the host's electron estimate is run a second time with the general sequence (wayne_ctx_reruns counts
 * them), an overflow is reported as WAYNE_E_OVERFLOW.  After WAYNE_OK the reads of every slot are complete: this is
 * the call a throughput loop of wayne_exposure_run ends with. */
int wayne_ctx_synchronize(wayne_ctx *ctx);
/* Tuning and test knobs.  Each is read ONCE from the environment (WAYNE_<NAME IN CAPITALS>) by wayne_ctx_create; no
 * other entry point looks at the environment, so editing it from another thread cannot change what a live context
 * launches.  Afterwards only this call changes a knob (value < 0: back to the library's own choice).  Names:
 * tile_ints, batch, thin, no_acc_box, lane_reach, throw_wgs, keep_narrow, no_fuse, fork_narrow, streams,
 * upload_timing, ramp_reads (timing builds only).  None changes a frame (integer accumulation commutes); they change
 * launch shapes.  WAYNE_E_INVALID for an unknown name. */
int wayne_ctx_set_knob(wayne_ctx *ctx, const char *name, long long value);
int wayne_ctx_get_knob(const wayne_ctx *ctx, const char *name, long long *value);
/* The context's hipStream_t (as void*), for callers that interoperate. */
void *wayne_ctx_stream(wayne_ctx *ctx);

/* ---- inner boundary: the electron thrower ------------------------------ */

/*
 * Drop-in for  int *PSF(counts,size,x_pos,y_pos,psf_ratio,psf_sigmal,
 *                       psf_sigmah,nr,nc,test,threads)
 * (wayne/pyparallel_menu.h:1-3, pyparallel_menu.c:10-113) and therefore for
 * wayne.pyparallel.apply_psf (wayne/pyparallel.pyx:14-38), called once per
 * sub-sample at exposure_generator.py:636-639.
 *
 * Differences from the reference signature: the frame is written into the
 * caller's `out` (nr*nc int32) instead of a malloc'd buffer the caller must
 * free (pyparallel.pyx:36); `seed` is the reference's `test`;
 * `threads_compat` selects the reference's OpenMP partition in replay mode
 * (the result depends on it, pyparallel_menu.c:47-52) and never starts CPU
 * threads; in Philox mode `exposure`/`subsample` extend the counter.
 * WAYNE_RNG_REPLAY reproduces the reference frame bit for bit.
 */
int wayne_psf_apply(wayne_ctx *ctx, const int32_t *counts, int size,
                    const double *x_pos, const double *y_pos,
                    const double *psf_ratio, const double *psf_sigmal,
                    const double *psf_sigmah, int nr, int nc, uint32_t seed,
                    int threads_compat, int rng_mode, uint32_t exposure,
                    uint32_t subsample, int32_t *out);
/* The same with `flags`: WAYNE_F_EXACT_SAMPLERS runs the split mode's binomial chains (k_narrow) with IEEE
 * divide and libm-grade exp / log instead of the hardware approximations -- same algorithm, same streams; for
 * parity runs against oracle/split_oracle.c.  wayne_psf_apply is this call with flags = 0 (production math). */
int wayne_psf_apply_ex(wayne_ctx *ctx, const int32_t *counts, int size,
                       const double *x_pos, const double *y_pos,
                       const double *psf_ratio, const double *psf_sigmal,
                       const double *psf_sigmah, int nr, int nc, uint32_t seed,
                       int threads_compat, int rng_mode, uint32_t exposure,
                       uint32_t subsample, uint32_t flags, int32_t *out);

/* ---- grism and calibration (uploaded once per context) ----------------- */

/* What grism.G141 / grism.G102 hold (grism.py:24-118, 426-476, 756-776). */
typedef struct wayne_grism_desc {
  double trace_coeff[9];     /* aXe trace polynomial   (grism.py:756-764)   */
  double wl_solution[9];     /* aXe dispersion solution (grism.py:768-776)  */
  double psf_ratio_poly[4];  /* np.poly1d coefficients, highest power first */
  double psf_sigmal_poly[4]; /*   (grism.py:85-90)                          */
  double psf_sigmah_poly[4];
  int n_sens;                /* sensitivity table (grism.py:97-106)         */
  const double *sens_wl_um;  /* micron; finite and non-decreasing (np.interp's precondition), */
  const double *sens_val;    /* finite values: otherwise WAYNE_E_INVALID and nothing changes     */
  double flat_wmin, flat_wmax; /* WMIN / WMAX of the flat cube (grism.py:71-72), angstrom */
} wayne_grism_desc;

int wayne_ctx_set_grism(wayne_ctx *ctx, const wayne_grism_desc *g);

/*
 * Calibration planes for one (SUBARRAY, SAMPSEQ, NSAMP) mode, already
 * centre-cropped to the sub-array by the caller (tools.crop_central_box,
 * tools.py:317-324; detector.py:328-333).  float32 as in the CALWF3 files.
 * Any pointer may be NULL when the matching flag is never used.
 */
typedef struct wayne_calibration {
  int subarray;          /* 64, 128, 256, 512 or 1024                          */
  int n_reads;           /* R = NSAMP - 1 non-zero reads                       */
  const float *flat[4];  /* N*N each: flat cube planes f0..f3 (grism.py:73-76) */
  const float *pfl;      /* N*N: pixel flat; gain = 2.35 / pfl (detector.py:200-209) */
  const float *sky;      /* N*N: master sky (grism.py:411-423)                 */
  const float *lin[4];   /* S*S each: non-linearity c1..c4 (detector.py:58-67) */
  const float *dark_sci; /* R*S*S: super-dark SCI of non-zero read r (detector.py:185-188) */
  const float *dark_err; /* R*S*S: its ERR                                      */
  const double *zero_read; /* S*S initial bias or NULL (exposure_generator.py:446-466) */
} wayne_calibration;

int wayne_ctx_set_calibration(wayne_ctx *ctx, const wayne_calibration *c);

/* ---- outer boundary: one whole exposure -------------------------------- */

/*
 * Everything ExposureGenerator.scanning_frame / staring_frame
 * (exposure_generator.py:146-405) consumes for one exposure, as plain arrays.
 * The sample timing (A12), scan positions, jitter and SSV scaling are small
 * K-vectors prepared by the host (wayne_amd/exposure_generator.py); the
 * per-wavelength, per-electron and per-pixel work happens on the device:
 *   trace + counts chain   exposure_generator.py:581-634, grism.py:491-669, 779-803
 *   electron thrower       pyparallel_menu.c:10-113
 *   flat                   grism.py:349-409, exposure_generator.py:641-645
 *   per-read stage         exposure_generator.py:468-515, cosmic_rays.py:70-139
 *   post-ramp stage        exposure_generator.py:407-444, exposure.py:49-131,
 *                          detector.py:151-198, 318-350
 */
typedef struct wayne_exposure_desc {
  uint32_t seed;           /* visit seed (run_visit.py:68-77)                  */
  uint32_t exposure_index; /* extends every RNG counter                        */
  int rng_mode;            /* thrower RNG: WAYNE_RNG_*                         */
  int threads_compat;      /* replay mode only                                 */
  uint32_t flags;          /* WAYNE_F_*                                        */
  int sub_scale;           /* frame offset 507 - SUBARRAY/2 (exposure_generator.py:630) */

  int n_wl;                /* W: bins already cropped to grism.wl_limits       */
  const double *wl_um;     /* [W] increasing                                   */
  const double *flux;      /* [W] stellar flux                                 */
  const double *depth;     /* [K*W] transit depth per sub-sample, or NULL      */

  int n_samples;              /* K                                             */
  const double *x_ref;        /* [K] star x incl. jitter                       */
  const double *y_ref;        /* [K] star y incl. scan + jitter                */
  const double *dur_ms;       /* [K] sub-sample duration (after SSV)           */
  const int32_t *replay_seed; /* [K] s_rand_seeds (exposure_generator.py:327) */
  const int32_t *sample_read; /* [K] index 0..R-1 of the read that closes it   */

  int n_reads;              /* R                                               */
  const double *read_dt_s;  /* [R] interval since the previous read, seconds   */

  double sky_ct_s;     /* sky background counts/s; <= 0 disables               */
  double cosmic_rate;  /* hits/s per 1024^2; < 0 disables (None)               */
  double scale_factor; /* visit-trend scale (1 when None)                      */
  double noise_mean;   /* optional gaussian noise per second, both 0 disables  */
  double noise_std;

  int thrower_margin; /* LDS tile margin in px around the trace; 0 = default   */
  int thrower_splits; /* thrower workgroups launched per sub-sample (an upper bound: the kernel
                         shares the electrons it finds among as many as it needs); 0 = sized
                         from the host's estimate of the electron count           */

  /* Device light curves (replaces the W pylightcurve calls per exposure of
   * Observation.generate_lightcurves, observation.py:293-357).  When lc_z is
   * not NULL, `depth` must be NULL and the K x W transit-depth matrix is
   * computed on the device by k_lightcurve:
   *   depth[k][w] = (1 - transit(z_k, rp_w, ld)) + (1 - eclipse(rp_w^2, hidden_k))      */
  const double *lc_z;      /* [K] projected separation in stellar radii (>= 10: planet behind the star) */
  const double *lc_hidden; /* [K] fraction of the planet's disk hidden by the star (eclipse), or NULL   */
  const double *lc_rp;     /* [W] Rp/R* per bin = sqrt(planet_spectrum)                                  */
  double lc_ld[4];         /* Claret 4-coefficient limb darkening                                        */
} wayne_exposure_desc;

/* Stage an exposure's inputs in HBM slot `slot` (0 <= slot < wayne_ctx_slots). */
int wayne_ctx_slots(const wayne_ctx *ctx);
int wayne_exposure_upload(wayne_ctx *ctx, int slot, const wayne_exposure_desc *d);
/* Enqueue the whole synthesis of slot `slot` on the context stream
 * (asynchronous; inputs and outputs stay in HBM). */
int wayne_exposure_run(wayne_ctx *ctx, int slot);
/* wayne_exposure_run, then wait for the slot and settle it (see wayne_ctx_synchronize): when this returns WAYNE_OK the
 * slot's reads in HBM are complete.  The blocking form for callers that read wayne_exposure_device_reads directly. */
int wayne_exposure_run_checked(wayne_ctx *ctx, int slot);
/* The status word of the slot's last run (synchronises its stream): 0 = complete; bit 0 = a count overflowed (the
 * download / wait / synchronize calls report it as WAYNE_E_OVERFLOW); bit 1 = a bin held more electrons than the
 * launch sequence chosen from the host's estimate handles.  Every call that hands results over or ends a batch --
 * wayne_exposure_download / _wait / _run_checked / _debug_fetch and wayne_ctx_synchronize -- looks at the word itself
 * and runs such an exposure a second time with the general sequence; only a caller that waits on wayne_ctx_stream
 * with its own HIP calls bypasses that and must look here.  wayne_ctx_reruns: how many such second runs the context
 * has made (their electrons are counted twice in wayne_profile.electrons). */
int wayne_exposure_status(wayne_ctx *ctx, int slot, int *status);
unsigned long long wayne_ctx_reruns(const wayne_ctx *ctx);
/* Copy the NSAMP reads (read 0 = zero read) of `slot` to the host:
 * NSAMP*S*S float32, or float64 when WAYNE_F_OUT_F64 was set.  Synchronises. */
int wayne_exposure_download(wayne_ctx *ctx, int slot, void *out_reads);
/* Pinned-host delivery for pipelines: wayne_exposure_fetch_async enqueues, on the
 * slot's stream (i.e. after its kernels), the copy of the reads into a pinned host
 * buffer owned by the library and returns at once; wayne_exposure_wait blocks until
 * that slot's work is done and returns the buffer (NSAMP*S*S float32 / float64), which
 * stays valid until the slot is uploaded again.  With two slots on the two streams the
 * copy of one exposure overlaps the kernels of the next. */
int wayne_exposure_fetch_async(wayne_ctx *ctx, int slot);
int wayne_exposure_wait(wayne_ctx *ctx, int slot, void **host_reads);
/* Device pointer of that buffer (for zero-copy consumers). */
void *wayne_exposure_device_reads(wayne_ctx *ctx, int slot);
/* upload + run + download in one call: the batched drop-in for one
 * ExposureGenerator.scanning_frame call. */
int wayne_exposure_synthesize(wayne_ctx *ctx, const wayne_exposure_desc *d,
                              void *out_reads);

/* Intermediate products of the last run of `slot`, for parity tests
 * (any pointer may be NULL):
 *   counts  [K*W] int32  electrons per bin per sub-sample   (A9)
 *   x_pos   [K*W] double frame x of each bin                (A7, A10)
 *   y_pos   [K*W] double
 *   acc_e   [R*S*S] double  flat-weighted electrons accumulated per read
 *                    interval, before sky / gain (A11, A12); includes cosmic hits
 * `acc_e` is only meaningful between wayne_exposure_run_front and
 * wayne_exposure_run_back (the ramp kernel clears it). */
int wayne_exposure_debug_fetch(wayne_ctx *ctx, int slot, int32_t *counts,
                               double *x_pos, double *y_pos, double *acc_e);
/* The K*W transit-depth matrix of `slot` (as uploaded, or as computed by k_lightcurve
 * in the last run_front), for parity tests. */
int wayne_exposure_debug_depth(wayne_ctx *ctx, int slot, double *depth);
/* Which accumulators the ramp kernel of `slot` loads (for the byte accounting of bench.py): boxes[r*4 .. r*4+3] =
 * {x0, x1, y0, y1} of read interval r in bordered coordinates, the host's bound on where the thrower's electrons of
 * that interval can land (all zero: everything is loaded), and segments[r] = the number of 64-accumulator segments
 * (one per wave) that intersect it.  Segments with a cosmic-ray hit are loaded too and not counted here.
 * boxes: 16*4 ints, segments: 16 ints.  Returns WAYNE_OK, *use_box = 0 when the slot loads every accumulator. */
int wayne_exposure_debug_boxes(wayne_ctx *ctx, int slot, int32_t *boxes, int32_t *segments, int *use_box);
/* The two halves of wayne_exposure_run: front = prep + thrower + cosmic rays,
 * back = the fused up-the-ramp kernel. */
int wayne_exposure_run_front(wayne_ctx *ctx, int slot);
int wayne_exposure_run_back(wayne_ctx *ctx, int slot);
/* Which instantiation of the fused up-the-ramp kernel the back half of `slot` launches, written into buf (cap bytes,
 * NUL-terminated) in the form a kernel trace prints it, e.g. "k_ramp<float, true, 1, false, true>": <type of the reads,
 * production (hardware) math, sky sampler 0 direct / 1 alias tables / 2 tables + pieces, gaussian-noise stage, every
 * detector switch on>.  The arithmetic differs between them (float: exact integer sums + an all-float32 per-read
 * chain; double: fp64 cumulative sum), so a measurement names the one it timed (bench.py `dtype`, `roofline.kernel`).
 * No reference counterpart: the reference has one float64 numpy path (exposure_generator.py:407-515). */
int wayne_exposure_ramp_variant(wayne_ctx *ctx, int slot, char *buf, int cap);

/* ---- measurement ------------------------------------------------------- */

#define WAYNE_PROF_KERNELS 8
typedef struct wayne_profile {
  /* per kernel: launches and total milliseconds measured with HIP events on
   * the context stream since wayne_profile_reset */
  const char *name[WAYNE_PROF_KERNELS];
  uint64_t launches[WAYNE_PROF_KERNELS];
  double ms[WAYNE_PROF_KERNELS];
  uint64_t electrons; /* electrons thrown */
} wayne_profile;

int wayne_profile_enable(wayne_ctx *ctx, int on);
/* Restrict the HIP-event timing to the kernels whose bit (1 << index in `name`) is set; every event
 * pair costs a few microseconds of stream time, so a throughput measurement that only needs one
 * kernel's duration selects that kernel.  Default: all kernels. */
int wayne_profile_select(wayne_ctx *ctx, unsigned mask);
int wayne_profile_reset(wayne_ctx *ctx);
int wayne_profile_get(wayne_ctx *ctx, wayne_profile *out); /* synchronises */

/* ---- host helpers ------------------------------------------------------ */

/* Philox4x32-10 block (Random123), used by the host for the per-exposure
 * jitter / seed draws (exposure_generator.py:327-329). */
void wayne_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The per-exposure host draws of scanning_frame (exposure_generator.py:327-329)
 * from the Philox HOST stage: for sub-sample k, block (k, 0, 0, exposure) gives
 * two standard normals (Box-Muller of words 0,1; the caller scales them by
 * x_jitter / y_jitter) and s_rand_seeds[k] = randint(0, 100000) from word 2.
 * Pure host arithmetic; any output pointer may be NULL. */
void wayne_host_sample_draws(uint32_t seed, uint32_t exposure, int n_samples,
                             double *z_x, double *z_y, int32_t *rand_seed);

#ifdef __cplusplus
}
#endif
#endif /* WAYNE_HIP_H */
