// libfrbch compute side: handle life cycle, kernel sequencing, rescale-interval state machine,
// host streaming (push/pull/run_file) and the device-resident entry points of include/frbch.h.
//
// Compiled as HIP for gfx950 (FRBCH_DEV_HEADER = "dev_hip.h").  The CPU unit tests compile the
// same file against tests/emu/dev_emu.h to exercise the host logic without a GPU; that build is
// test infrastructure and is never loaded by the package.
#include FRBCH_DEV_HEADER

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <sstream>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "frbch_host.h"
#include "frbch_kparams.h"

#include "kernels_generic.inc"
#ifndef FRBCH_NO_FAST
#include "kernels_fast.inc"
#include "kernels_k2priv.inc"
#endif

using namespace frbch;

namespace {

enum { KID_K1 = 0, KID_KC, KID_K2, KID_STATS, KID_QUANT, KID_K3, KID_K4, KID_K0, KID_K2S, KID_COUNT };
const char* const kKernelNames[KID_COUNT] = {"frbch_k1_branch", "frbch_kc_dcfix", "frbch_k2_chan",
                                             "frbch_stats", "frbch_quantise", "frbch_k3_dedisp", "frbch_k4_out", "frbch_k0_stage",
                                             "frbch_k2_statpass"};
static_assert(KID_COUNT <= (int)(sizeof(((frbch_timing*)nullptr)->k) / sizeof(((frbch_timing*)nullptr)->k[0])), "frbch_timing holds every slot");

struct EventPair {
  dev_event_t a, b;
  int kid;
  double bytes;
};

}  // namespace

struct frbch_handle {
  frbch_config cfg;
  Plan pl;
  std::string err;
  int device = 0;
  dev_stream_t stream = 0;
  dev_stream_t user_stream = 0;   // last caller stream the device entry points launched on (0 = none pending); never dereferenced
  dev_event_t user_ev{};          // ... and the handle's own event recorded on it behind that work
  bool user_ev_made = false;
  size_t lds_limit = 65536;

  // constant tables
  cf *tw_r = nullptr, *tw_c2 = nullptr, *tw_nhi = nullptr, *tw_nlo = nullptr;
  cf *ftw1_r = nullptr, *ftw2_r = nullptr, *ftw1_c = nullptr, *ftw2_c = nullptr, *td1 = nullptr, *td2 = nullptr;
  cf *ftw1_h = nullptr, *ftw2_h = nullptr;
  // per-launch work buffers
  cf *spill = nullptr, *s_dc = nullptr, *p0 = nullptr;
  // coherent dedispersion (-F C:D): second spill, kernel table, channel-major power
  cf *spill2 = nullptr, *chirp = nullptr;
  int coh_order_m = 0;         // order of the fine bins in spill2 / chirp: 0 bit-reversed (generic K1/K3), M: register passes
  float* ptmp = nullptr;
  // rescale state
  float *offset = nullptr, *scale = nullptr;
  bool have_scale = false;     // offset/scale are defined
  bool scale_frozen = false;   // ... and stay as they are (set_rescale, -c after 1st interval, -I0)
  float* powbuf = nullptr;     // float power of the interval being measured [row][ncol]
  float* scr2 = nullptr;       // two-stage tscrunch (Plan::k2_two_stage): [maxb][R/2][C] rows of two time samples
  uint64_t pow_cap_rows = 0, pow_rows = 0;
  double* partial = nullptr;
  int partial_chunks = 0;
  // rescale statistics accumulated by the fast K2 while it emits the float power (no second pass)
  uint64_t fused_rows = 0;     // rows of the current interval whose moments are in `partial`
  bool fused_valid = false;    // ... and no row of the interval is missing from them
  int fused_chunks = 0;        // rows of `partial` the fused path uses (0 = this configuration cannot fuse)

  uint64_t rows_out = 0, blocks_done = 0;
  // two-pass rescale: the first batch of a `-c` rescale interval is not written as float rows.  K2 runs over
  // the resident spill twice instead -- a statistics-only pass at once, the digitising pass when the interval is complete (inside
  // the batch, or at the flush).  Until then the batch is DEFERRED: its spill, S and dP stay untouched.
  struct Deferred {
    bool active = false;
    KParams p;            // the batch's launch parameters as launch_front left them
    uint32_t nb = 0;
    uint64_t rows = 0;
  } deferred;
  int priv_grid = 0;              // workgroups of frbch_k2_priv (one per CU): its rows of partial rescale sums

  // VDIF stream state (host streaming path)
  bool have_vdif = false;
  VdifInfo v0{};
  double tstart_mjd = 0.0;
  uint64_t skip_bytes = 0;      // payload bytes still to skip before the next block starts
  uint64_t blocks_budget = 0;   // blocks still allowed by -T
  std::vector<uint8_t> carry;   // whole + partial frames not yet consumed
  uint64_t frames_seen = 0, frames_invalid = 0, frame_gaps = 0, frames_filled = 0;
  std::vector<uint8_t> carry_bad;  // per frame of `carry`: 1 = flagged invalid, or a filler inserted for a missing frame number
  uint32_t* d_fbad = nullptr;      // bitmap of those flags for the frames of the launch in progress
  size_t d_fbad_words = 0;
  uint64_t next_frame_index = 0;   // seconds*fps + frame_nr expected next
  size_t checked_bytes = 0;        // prefix of `carry` whose headers were already checked
  uint8_t* d_frames = nullptr;
  size_t d_frames_cap = 0;
  uint8_t* d_out = nullptr;
  size_t d_out_cap = 0;
  uint8_t* stg = nullptr;          // corner-turned payload of one launch batch for the wave K1 (frbch_k0_stage)
  bool stg_ready = false;          // ... valid for the launch in progress
  std::vector<uint8_t> outq;
  size_t outq_pos = 0;
  // whole-file paths: pinned host rings, kept for the life of the handle (pinning costs ~0.5 ms per MB)
  uint8_t* pin_in[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t* pin_out[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t pin_in_cap = 0, pin_out_cap = 0;
  // frbch_run_scan: rows go into this IF's columns of a pitched device buffer shared by the IFs of the scan
  uint8_t* sink = nullptr;         // first byte of this IF's columns in line 0
  size_t sink_line_pitch = 0;      // bytes per (row, product) line of the shared buffer
  uint64_t sink_rows = 0, sink_rows_cap = 0;

  // rows for a wider row buffer: values between consecutive (row, product) lines of code_out (0 = nchan: packed rows)
  uint64_t out_pitch = 0;
  // two-lane pipeline: spill regions in flight, ordering events
  uint8_t* stg_cur = nullptr;      // staged payload of the launch in progress (stg + region offset)
  int lane_cus = 0;                // compute units of the stream the next K1 goes to (0 = all of them)
  int lane_ncu = 0;                // compute units of the device (frbch_open)
  bool quant_lds_allowed = false;  // the digitiser's LDS reservation (overlap mode 3) was enabled on this handle's device
  dev_event_t region_ev[8];        // recorded behind the last back stage that read spill region r
  bool region_busy[8] = {false, false, false, false, false, false, false, false};
  bool region_ev_made = false;
  std::vector<dev_event_t> evpool; // ordering events, used round-robin
  size_t evnext = 0;
  uint32_t next_region = 0;        // spill region of the next batch
  uint32_t diag = 0;               // frbch_info::diag
  uint8_t* scan_rows = nullptr;    // frbch_run_scan (first handle of the scan): the row buffer of all its IFs, kept between calls
  size_t scan_rows_bytes = 0;
  dev_event_t reset_ev{};          // behind the identity rescale a reset queued on the handle's stream
  bool reset_ev_made = false, reset_pending = false;
  dev_event_t quant_ev{};          // behind a digitiser that ran on the back lane (mode 2)
  bool quant_ev_made = false, quant_busy = false;
  int quant_lane_cus = 0;          // CUs of the lane the digitiser was sent to

  // profiling
  bool profiling = false;
  std::vector<EventPair> events;
  std::string kname[KID_COUNT];   // kernel actually launched in each slot (for the timing report)
  double acc_ms[KID_COUNT] = {0};
  double acc_bytes[KID_COUNT] = {0};
  uint64_t acc_launches[KID_COUNT] = {0};
};

namespace {

int fail(frbch_handle* h, int code, const std::string& msg) {
  h->err = msg;
  return code;
}

// Every entry point that touches HIP runs with the handle's device current and restores the caller's device on the
// way out: handles may be driven from any thread, next to other handles or to torch on other GPUs.
struct DeviceGuard {
  int prev, want;
  explicit DeviceGuard(int dev) : prev(dev_get()), want(dev) {
    if (prev != want) (void)dev_set(want);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != want) (void)dev_set(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// cfg.flags of a product build (include/frbch.h): four kernel-selection switches, every one produces the same (correct) output
// and has parity cases.  Everything else -- rejected kernel variants, layouts and lane modes kept for A/B runs, the timing-only
// ablations of bits 8..19 -- exists only in libraries built with -DFRBCH_EXPERIMENTS; bit 22 (whole-file paths without their
// reader / writer threads) also in the test-only emulator build.
constexpr uint32_t kFlagGenericK1 = 1u, kFlagGenericK2 = 2u, kFlagSeparateStats = 1u << 20, kFlagTwoPass = 1u << 27;
constexpr uint32_t kProductFlags = kFlagGenericK1 | kFlagGenericK2 | kFlagSeparateStats | kFlagTwoPass;
constexpr uint32_t kFlagNoPipeline = 1u << 22, kFlagNoK0 = 1u << 23, kFlagGenericQuant = 1u << 25;
[[maybe_unused]] constexpr uint32_t kExperimentFlags = 4u | 8u | 16u | 32u | 64u | 128u | (1u << 21) | (1u << 22) | (1u << 23) | (1u << 24) | (1u << 25) | (1u << 26);
#ifdef FRBCH_EXPERIMENTS
constexpr bool kExperiments = true;
constexpr uint32_t kAcceptedFlags = kProductFlags | kExperimentFlags | 0x000FFF00u;
#elif defined(FRBCH_TEST_HOOKS)
constexpr bool kExperiments = false;
constexpr uint32_t kAcceptedFlags = kProductFlags | kFlagNoPipeline;
#else
constexpr bool kExperiments = false;
constexpr uint32_t kAcceptedFlags = kProductFlags;
#endif

#define CHECK_DEV(h, expr, what)                                                           \
  do {                                                                                     \
    if ((expr) != 0) return fail((h), FRBCH_E_DEVICE, std::string(what) + ": " + dev_last_error_string()); \
  } while (0)

struct ProfScope {
  frbch_handle* h;
  dev_stream_t s;
  EventPair ep;
  bool on;
  ProfScope(frbch_handle* h_, dev_stream_t s_, int kid, double bytes) : h(h_), s(s_), on(h_->profiling) {
    if (on) {
      ep.kid = kid;
      ep.bytes = bytes;
      dev_event_create(&ep.a);
      dev_event_create(&ep.b);
      dev_event_record(ep.a, s);
    }
  }
  ~ProfScope() {
    if (on) {
      dev_event_record(ep.b, s);
      h->events.push_back(ep);
    }
  }
};

void drain_events(frbch_handle* h) {
  for (auto& e : h->events) {
    h->acc_ms[e.kid] += dev_event_ms(e.a, e.b);
    h->acc_bytes[e.kid] += e.bytes;
    h->acc_launches[e.kid] += 1;
    dev_event_destroy(e.a);
    dev_event_destroy(e.b);
  }
  h->events.clear();
}

int upload_table(frbch_handle* h, cf** dst, uint64_t n, uint64_t count, uint64_t step) {
  std::vector<float> tmp(2 * count);
  fill_twiddles(tmp.data(), n, count, step);
  CHECK_DEV(h, dev_malloc((void**)dst, count * sizeof(cf)), "hipMalloc(twiddles)");
  CHECK_DEV(h, dev_h2d(*dst, tmp.data(), count * sizeof(cf), h->stream), "upload twiddles");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  return FRBCH_OK;
}

KParams base_params(const frbch_handle* h) {
  const Plan& pl = h->pl;
  KParams p;
  memset(&p, 0, sizeof p);
  p.log2_c2 = pl.log2_c2;
  p.log2_r = pl.log2_r;
  p.c = pl.c;
  p.c2 = pl.c2;
  p.r = pl.r;
  p.g = pl.g;
  p.log2_g = 0;
  while ((1 << p.log2_g) < pl.g) ++p.log2_g;
  p.tt = pl.tt;
  p.tscr = pl.tscr;
  p.nif = pl.nif;
  p.pol_mode = h->cfg.pol_mode;
  p.nbit = h->cfg.nbit_out;
  p.flip = pl.flip;
  p.log2_nlo = pl.log2_nlo;
  p.in_bits = pl.in_bits;
  p.spill = h->spill;
  p.gs = pl.gs;
  p.s_dc = h->s_dc;
  p.p0 = h->p0;
  p.tw_r = h->tw_r;
  p.tw_c2 = h->tw_c2;
  p.tw_nhi = h->tw_nhi;
  p.tw_nlo = h->tw_nlo;
  p.ftw1_r = h->ftw1_r;
  p.ftw2_r = h->ftw2_r;
  p.ftw1_h = h->ftw1_h;
  p.ftw2_h = h->ftw2_h;
  p.ftw1_c = h->ftw1_c;
  p.ftw2_c = h->ftw2_c;
  p.td1 = h->td1;
  p.td2 = h->td2;
  p.offset = h->offset;
  p.scale = h->scale;
  {   // 2-bit level table: DSPSR's static one unless the configuration brings its own
    const float* lv = h->cfg.levels;
    const bool own = lv[0] != 0.f || lv[1] != 0.f || lv[2] != 0.f || lv[3] != 0.f;
    static const float dflt[4] = {-3.3359f, -1.0f, 1.0f, 3.3359f};
    for (int i = 0; i < 4; ++i) p.lut[i] = own ? lv[i] : dflt[i];
  }
  p.digi_mean = pl.digi_mean;
  p.digi_scale = pl.digi_scale;
  p.digi_max = pl.digi_max;
  p.out_pitch = h->out_pitch ? h->out_pitch : (uint64_t)pl.c;
#ifdef FRBCH_EXPERIMENTS
  p.dbg = (h->cfg.flags >> 8) & 0xFFFu;   // bits 8..19: timing-only ablations (wrong output)
#else
  p.dbg = 0;
#endif
  p.coherent = pl.coherent;
  p.nfilt_pos = pl.nfilt_pos;
  p.keep = pl.keep;
  p.hop = pl.hop;
  p.spill2 = h->spill2;
  p.chirp = h->chirp;
  p.ptmp = h->ptmp;
  return p;
}

#ifndef FRBCH_NO_FAST
template <int LOG2M>
void launch_k1_fast_t(const Plan& pl, KParams& p, uint32_t nb, dev_stream_t s) {
  hipLaunchKernelGGL(fast::frbch_k1_fast<LOG2M>, dim3(pl.c2 / pl.g, nb), dim3(1024), pl.k1_fast_lds, s, p);
}
template <int LOG2M>
void launch_k1_wave_t(const Plan& pl, KParams& p, uint32_t nb, dev_stream_t s, int ncu) {
  // persistent over blocks: the resident workgroups each keep their branch group and loop over the batch
  p.nblk = nb;
#ifdef FRBCH_EXPERIMENTS
  static const int stag_env = getenv("FRBCH_K1_STAG") ? atoi(getenv("FRBCH_K1_STAG")) : 3;   // priority schedule (3: the halves swap priority behind the forward passes; measured 1.96 -> 1.94 ms)
  p.stag = stag_env;
#else
  p.stag = 3;
#endif
  const int kg = pl.fast_k1_g;            // branches per workgroup (<= pl.g, the layout group)
  {
    static const int ks[6] = {1, 2, 3, 4, 8, 12};
    const int step = 64 / kg;
    for (int i = 0; i < 6; ++i) {
      const double a = -2.0 * M_PI * (double)((step * ks[i]) % pl.r) / (double)pl.r;
      p.rot6[i].x = (float)cos(a);
      p.rot6[i].y = (float)sin(a);
    }
  }
  const uint32_t ngrp = (uint32_t)(pl.c2 / kg);
#ifdef FRBCH_EXPERIMENTS
  static const char* stamp_path = getenv("FRBCH_STAMPS");   // diagnostic: phase stamps of one block, dumped after every launch
#else
  const char* const stamp_path = nullptr;
#endif
  static unsigned long long* stamp_buf = nullptr;
  const size_t stamp_n = (size_t)ngrp * 16 * 16;
  if (stamp_path) {
    if (!stamp_buf) (void)hipMalloc((void**)&stamp_buf, (size_t)4096 * 16 * 16 * 8);
    (void)hipMemsetAsync(stamp_buf, 0, stamp_n * 8, s);
    p.stamps = stamp_buf;
  }
#ifdef FRBCH_EXPERIMENTS
  static const uint32_t cap_env = getenv("FRBCH_K1_MAXWG") ? (uint32_t)atoi(getenv("FRBCH_K1_MAXWG")) : 0u;
#else
  const uint32_t cap_env = 0u;
#endif
  const uint32_t resident = cap_env ? cap_env : (uint32_t)(ncu > 0 ? ncu : 256) * (uint32_t)std::max<size_t>(1, (160 * 1024) / pl.k1_fast_lds);
  uint32_t ny = std::max<uint32_t>(1, std::min<uint32_t>(nb, resident / std::max<uint32_t>(1, ngrp)));
  if (ngrp > resident && ngrp % resident != 0) {
    // more branch groups than resident workgroups and not a whole number of rounds (a CU-masked lane, e.g. 256 groups on
    // 160 CUs): split the blocks over ny workgroups per group so that ngrp * ny fills whole rounds
    uint32_t g = ngrp, r = resident;
    while (r) { const uint32_t t = g % r; g = r; r = t; }
    uint32_t want = resident / g;
#ifdef FRBCH_EXPERIMENTS
    static const int ny_env = getenv("FRBCH_K1_NY") ? atoi(getenv("FRBCH_K1_NY")) : 0;
    if (ny_env > 0) want = (uint32_t)ny_env;
#endif
    if (want <= nb) ny = want;
  }
#define FRBCH_K1W(L, NWV, WPSV, NTV)                                                                                       \
  do {                                                                                                                  \
    if (p.stg) hipLaunchKernelGGL((fast::frbch_k1_wave<L, NWV, WPSV, true>), dim3(ngrp, ny), dim3(NTV), pl.k1_fast_lds, s, p);  \
    else hipLaunchKernelGGL((fast::frbch_k1_wave<L, NWV, WPSV, false>), dim3(ngrp, ny), dim3(NTV), pl.k1_fast_lds, s, p);        \
  } while (0)
  if constexpr (LOG2M == 5) {
    FRBCH_K1W(5, 8, 4, 512);      // R = 8192: two branches per workgroup, four waves (two virtual threads per lane) each
  } else if constexpr (LOG2M == 4) {
    if (p.coherent) {   // forward transform + delay only, spectrum spilled (K2c follows)
      if (p.stg) hipLaunchKernelGGL((fast::frbch_k1_wave<4, 8, 2, true, true>), dim3(ngrp, ny), dim3(512), pl.k1_fast_lds, s, p);
      else hipLaunchKernelGGL((fast::frbch_k1_wave<4, 8, 2, false, true>), dim3(ngrp, ny), dim3(512), pl.k1_fast_lds, s, p);
    } else
    FRBCH_K1W(4, 8, 2, 512);
  } else {
#ifdef FRBCH_EXPERIMENTS   // K1 shapes with more, smaller waves (flags 64 / 128): all measured slower than eight wave-private waves
    if (LOG2M == 3 && pl.fast_k1_kind == 1) FRBCH_K1W(3, 4, 1, 256);
    else if (LOG2M == 3 && pl.fast_k1_kind == 2) FRBCH_K1W(3, 8, 2, 512);
    else if (LOG2M == 3 && pl.fast_k1_kind == 3) FRBCH_K1W(3, 16, 2, 1024);
    else
#endif
    FRBCH_K1W(LOG2M, 8, 1, 512);
  }
#undef FRBCH_K1W
  if (stamp_path && nb > 8 * ny) {
    std::vector<unsigned long long> hst(stamp_n);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(hst.data(), stamp_buf, stamp_n * 8, hipMemcpyDeviceToHost);
    if (FILE* f = fopen(stamp_path, "wb")) { fwrite(hst.data(), 8, stamp_n, f); fclose(f); }
  }
}
constexpr uint32_t kFusedStatWgs = 2048;   // persistent K2 workgroups (= rows of partial sums per thread row) while statistics are fused
// threads per workgroup of the wave-private K2 variant launch_k2_wave_t selects
int k2_wave_nt(const Plan& pl, uint32_t h_flags) {
  if (pl.fast_k2_log2m == 5) return 512;
  if (pl.fast_k2_log2m == 4) return pl.fast_k2_nw == 2 ? 256 : 512;
  if (pl.fast_k2_nw == 8) return 512;
  if (pl.fast_k2_log2m == 3 && !(h_flags & 32u)) return pl.fast_k2_nw == 2 ? 256 : 512;
  return pl.fast_k2_nw == 2 ? 128 : 256;
}
// rows of partial sums the fused statistics use; 0 = this configuration cannot fuse (one column group per thread needed)
// the wave K3 (coherent filterbank, R = 4096) sums the statistics of its channel: one row of partial sums per persistent workgroup
constexpr uint32_t kK3WaveWgs = 2048;
bool k3_wave_planned(const Plan& pl, uint32_t h_flags) {
  return pl.coherent && pl.coh_fast_r == 4 && pl.coh_nt == 512 && !(h_flags & 8u);
}
// frbch_k2_priv: one row of partial sums per (workgroup, row phase); `grid` = its workgroups (one per CU)
int priv_stat_chunks(const Plan& pl, int grid) { return grid * (pl.ncol / 4 >= 256 ? 1 : (int)(256 / (pl.ncol / 4))); }
// which K2 a launch of a plan with frbch_k2_priv takes: float rows of four products stay on frbch_k2_wave (measured, config 3:
// 1.87 ms per IF against 1.97 -- the phase is HBM-bound and the two-wave kernel reads whole 128-byte lines, frbch_k2_priv halves of
// them twice), everything else -- codes, one product, statistics only -- runs frbch_k2_priv (steady state of config 3 + 2.4 %,
// config 2 + 4 %)
bool pol_mode_no_sums(int pol_mode) { return pol_mode == 3; }   // (PP+QQ)^2: its square overflows the fp32 partial sums (~1e24 squared)
bool priv_takes(const Plan& pl, const KParams& p, int priv_grid) {
  // (the two-wave kernel reads the tile-major spill only in its two-sample form: fast_k2_nw == 2, tscrunch <= 2)
  return pl.fast_k2_priv && p.tile_major == 2 && priv_grid > 0 &&
         !(p.out_mode == FRBCH_OUT_FLOAT_POWER && pl.nif == 4 && pl.fast_k2_nw == 2 && pl.fast_k2_log2m == 3);
}
int wave_stat_chunks(const Plan& pl, uint32_t h_flags, int pol_mode);
// rows of the table of partial rescale sums the kernels of this plan may write (both K2 families add into the same table: whatever
// mix of them ran, frbch_stats_final sums every row)
int fused_stat_chunks(const Plan& pl, uint32_t h_flags, int pol_mode, int priv_grid = 0) {
  if (pol_mode == 3 || pl.k2_two_stage || (h_flags & (1u << 20))) return wave_stat_chunks(pl, h_flags, pol_mode);
  const int w = wave_stat_chunks(pl, h_flags, pol_mode);
  return (pl.fast_k2_priv && priv_grid > 0) ? std::max(w, priv_stat_chunks(pl, priv_grid)) : w;
}
int wave_stat_chunks(const Plan& pl, uint32_t h_flags, int pol_mode) {
  if (pol_mode == 3 || pl.k2_two_stage) return 0;   // (two-stage tscrunch: K2 does not see the output rows)
  if (k3_wave_planned(pl, h_flags)) return (h_flags & (1u << 20)) ? 0 : (int)kK3WaveWgs;   // (PP+QQ)^2: its square overflows the fp32 partial sums (~1e24 squared)
  if (!(pl.fast_k2_log2m || pl.fast_k2_m1) || !pl.fast_k2_wave || pl.coherent || (h_flags & (1u << 20))) return 0;
  const int nt = k2_wave_nt(pl, h_flags), cg = (int)(pl.ncol / 4);
  if (cg > nt)   // a thread owns cg/nt column groups, one row of sums per workgroup (the MSTAT instantiations: 2C = 2048, two waves per sequence)
    return (cg % nt == 0 && cg / nt <= 4)
               ? (pl.fast_k2_log2m == 5 ? 256 : ((pl.fast_k2_log2m == 3 && !(h_flags & 32u) && pl.fast_k2_nw != 8) ? (int)kFusedStatWgs : 0))   // 2C = 8192: resident workgroups only
               : 0;
  if (nt % cg != 0) return 0;
  return (int)kFusedStatWgs * (nt / cg);
}
template <int LOG2M>
void launch_k2_wave_t(const Plan& pl, KParams& p, uint32_t nb, dev_stream_t s, uint32_t h_flags) {
  const int tps = 16 << LOG2M;
  const int spw = tps < 64 ? 64 / tps : 1;
  // persistent: one wave of workgroups loops over the (tiles per block) x nb tiles of the launch
  p.nblk = nb;
  if (!wave_stat_chunks(pl, h_flags, p.pol_mode)) p.stat_partial = nullptr;
#ifdef FRBCH_EXPERIMENTS
  static const uint32_t npers_env = getenv("FRBCH_K2_NPERS") ? (uint32_t)atoi(getenv("FRBCH_K2_NPERS")) : 0u;
#else
  const uint32_t npers_env = 0u;
#endif
  const uint32_t npers = p.stat_partial ? kFusedStatWgs : (npers_env ? npers_env : 8192u);   // measured: 768 (= resident) 1.59 ms, 2048 1.56, 8192 1.49 (shorter tail)
  auto pers = [&](uint32_t tiles_per_block) { return dim3(std::min<uint64_t>((uint64_t)tiles_per_block * nb, npers)); };
  const dim3 grid2 = pers(pl.r / (2 * spw)), grid4 = pers(pl.r / (4 * spw)), grid8 = pers(pl.r / (8 * spw));
  const int pm = p.pol_mode == 2 ? 2 : (p.pol_mode >= 4 ? 4 : 0);
#define FRBCH_K2W(NWV, PMV, GRID) hipLaunchKernelGGL((fast::frbch_k2_wave<LOG2M, NWV, PMV>), GRID, dim3(64 * NWV), pl.k2_fast_lds, s, p)
  if constexpr (LOG2M == 5) {   // 2C = 8192: two time samples per workgroup (32-byte pieces of the spill lines), four waves and two virtual threads per lane each; one workgroup per CU
    const dim3 grid1 = dim3(std::min<uint64_t>((uint64_t)(pl.r / 2) * nb, p.stat_partial ? 256u : 1024u));   // (multiples of 8: XCD-aware tile order)
    // the instantiation with the per-thread column registers: the rescale sums while an interval is being measured, the
    // frozen offset / scale when it digitises (loaded in the emit they wait for the whole prefetch: 2.28 vs 2.1 ms);
    // the plain one for float rows without sums (two-stage tscrunch)
    const bool cols = p.stat_partial || p.out_mode != FRBCH_OUT_FLOAT_POWER;
    if (pm == 2 && cols) hipLaunchKernelGGL((fast::frbch_k2_wave<5, 8, 2, 4, true>), grid1, dim3(512), pl.k2_fast_lds, s, p);
    else if (pm == 2) hipLaunchKernelGGL((fast::frbch_k2_wave<5, 8, 2, 4>), grid1, dim3(512), pl.k2_fast_lds, s, p);
    else if (cols) hipLaunchKernelGGL((fast::frbch_k2_wave<5, 8, 0, 4, true>), grid1, dim3(512), pl.k2_fast_lds, s, p);
    else hipLaunchKernelGGL((fast::frbch_k2_wave<5, 8, 0, 4>), grid1, dim3(512), pl.k2_fast_lds, s, p);
  } else
  if constexpr (LOG2M == 4) {   // 2C = 4096: two waves per sequence; 2 or 4 sequences per workgroup
    if (pl.fast_k2_nw == 2) {
      if (pm == 2) hipLaunchKernelGGL((fast::frbch_k2_wave<4, 4, 2, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
      else if (pm == 4) hipLaunchKernelGGL((fast::frbch_k2_wave<4, 4, 4, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
      else hipLaunchKernelGGL((fast::frbch_k2_wave<4, 4, 0, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
    } else {
      if (pm == 2) hipLaunchKernelGGL((fast::frbch_k2_wave<4, 8, 2, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
      else if (pm == 4) hipLaunchKernelGGL((fast::frbch_k2_wave<4, 8, 4, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
      else hipLaunchKernelGGL((fast::frbch_k2_wave<4, 8, 0, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
    }
  } else {
  if (pl.fast_k2_nw == 8) {   // large tscrunch: 8 (x spw) sequences per workgroup, one wave per sequence
    if (pm == 2) FRBCH_K2W(8, 2, grid8); else if (pm == 4) FRBCH_K2W(8, 4, grid8); else FRBCH_K2W(8, 0, grid8);
  } else
  // M = 8: two waves per sequence (16 points per lane), 2 or 4 sequences per workgroup -> 16 waves per CU
  if (LOG2M == 3 && !(h_flags & 32u)) {
    if (pl.fast_k2_nw == 2) {
      if (pm == 2) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 4, 2, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
      else if (pm == 4 && (p.stat_partial || p.out_mode != FRBCH_OUT_FLOAT_POWER)) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 4, 4, 2, true>), grid2, dim3(256), pl.k2_fast_lds, s, p);
      else if (pm == 4) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 4, 4, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
      else hipLaunchKernelGGL((fast::frbch_k2_wave<3, 4, 0, 2>), grid2, dim3(256), pl.k2_fast_lds, s, p);
    } else {
      if (pm == 2) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 8, 2, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
      else if (pm == 4 && (p.stat_partial || p.out_mode != FRBCH_OUT_FLOAT_POWER)) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 8, 4, 2, true>), grid4, dim3(512), pl.k2_fast_lds, s, p);
      else if (pm == 4) hipLaunchKernelGGL((fast::frbch_k2_wave<3, 8, 4, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
      else hipLaunchKernelGGL((fast::frbch_k2_wave<3, 8, 0, 2>), grid4, dim3(512), pl.k2_fast_lds, s, p);
    }
  } else if constexpr (LOG2M != 3 || kExperiments) {   // (2C = 2048 with one wave per sequence and 2 / 4 waves: flag 32, experiments builds only)
  if (pl.fast_k2_nw == 2) {
    if (pm == 2) FRBCH_K2W(2, 2, grid2); else if (pm == 4) FRBCH_K2W(2, 4, grid2); else FRBCH_K2W(2, 0, grid2);
  } else {
    if (pm == 2) FRBCH_K2W(4, 2, grid4); else if (pm == 4) FRBCH_K2W(4, 4, grid4); else FRBCH_K2W(4, 0, grid4);
  }
  }
  }
#undef FRBCH_K2W
}
void set_fastdiv(KParams& p) {
  const uint32_t d = p.payload_bytes;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  p.div_magic = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  p.div_shift = l ? l - 1 : 0;
}
template <int LOG2M>
void launch_k2_fast_t(const Plan& pl, KParams& p, uint32_t nb, dev_stream_t s) {
  const int tps = 16 << LOG2M;
  // time samples per workgroup: its TT-sample tile, or (M = 32 only) tscrunch/TT tiles in a row
  const int tile_t = std::max(pl.fast_k2_nt / tps, pl.tscr);
  if (pl.fast_k2_nt == 1024)
    hipLaunchKernelGGL((fast::frbch_k2_fast<LOG2M, 1024>), dim3(pl.r / tile_t, nb), dim3(1024), pl.k2_fast_lds, s, p);
  else
    hipLaunchKernelGGL((fast::frbch_k2_fast<LOG2M, 512>), dim3(pl.r / tile_t, nb), dim3(512), pl.k2_fast_lds, s, p);
}
template <int LOG2M>
void launch_kc_fast_t(const Plan& pl, KParams& p, uint32_t nb, dev_stream_t s) {
  const size_t lds = ((size_t)pl.c2 + pl.c2 / 8 + 8 + pl.c2) * 8;
  hipLaunchKernelGGL(fast::frbch_kc_fast<LOG2M>, dim3(1, nb), dim3(16 << LOG2M), lds, s, p);
}
bool launch_kc_fast(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  if (pl.fast_k2_lane == 1) {   // 2C = 64: one lane per block
    p.nblk = nb;
    hipLaunchKernelGGL(fast::frbch_kc_lane, dim3((nb + 63) / 64), dim3(64), 0, s, p);
    return true;
  }
  switch (pl.fast_k2_log2m) {   // shares the 2C-point tables of the fast K2
    case 1: launch_kc_fast_t<1>(pl, p, nb, s); break;
    case 2: launch_kc_fast_t<2>(pl, p, nb, s); break;
    case 3: launch_kc_fast_t<3>(pl, p, nb, s); break;
    case 4: launch_kc_fast_t<4>(pl, p, nb, s); break;
    case 5: launch_kc_fast_t<5>(pl, p, nb, s); break;
    default: return false;
  }
  return true;
}
// corner-turn of the batch's payload for the wave K1 (own timing slot); same preconditions as launch_k1_fast
void launch_k0_stage(frbch_handle* h, const KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  h->stg_ready = false;
  const bool no_k0 = (h->cfg.flags & kFlagNoK0) != 0;   // gather straight from the frames
  if (!h->stg || no_k0 || !pl.fast_k1_log2m || pl.c % 256 != 0 || pl.r % 64 != 0) return;
  if (pl.coherent && !h->coh_order_m) return;              // the generic K1 is in use
  const uint32_t rb = (uint32_t)(pl.fast_k1_wave ? pl.fast_k1_g : pl.g) / 2;
  if (rb < 1 || p.payload_off % rb || p.payload_bytes % rb || p.header_bytes % rb || p.frame_bytes % rb || ((uintptr_t)p.frames % 16) ||
      p.payload_bytes < 2 || p.payload_off % 4 || p.payload_bytes % 4 || p.header_bytes % 4 || p.frame_bytes % 4)
    return;
  const uint64_t fr0 = p.payload_off / p.payload_bytes;
  const uint64_t rel0 = p.payload_off - fr0 * p.payload_bytes;
  if (rel0 + (uint64_t)(nb - 1) * pl.block_stride_bytes + pl.block_payload_bytes >= (1ull << 32)) return;
  KParams q = p;
  q.frames = p.frames + fr0 * p.frame_bytes;
  q.rel0 = (uint32_t)rel0;
  set_fastdiv(q);
  q.stg_out = h->stg_cur ? h->stg_cur : h->stg;
  ProfScope ps(h, s, KID_K0, (double)nb * (double)pl.block_payload_bytes * (1.0 + (double)p.frame_bytes / p.payload_bytes));
  const dim3 grid((pl.r / 64) * (pl.c / 256), nb);
  const bool wide = !(rel0 % 16 || p.payload_bytes % 16 || p.header_bytes % 16 || p.frame_bytes % 16);
#define FRBCH_K0(RBV) do { if (wide) hipLaunchKernelGGL((fast::frbch_k0_stage<RBV, true>), grid, dim3(256), 0, s, q); \
                           else hipLaunchKernelGGL((fast::frbch_k0_stage<RBV, false>), grid, dim3(256), 0, s, q); } while (0)
  switch (rb) {
    case 1: FRBCH_K0(1); break;
    case 2: FRBCH_K0(2); break;
    case 4: FRBCH_K0(4); break;
    case 8: FRBCH_K0(8); break;
    default: FRBCH_K0(16); break;
  }
#undef FRBCH_K0
  h->stg_ready = true;
}
bool launch_k1_fast(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  if (!pl.fast_k1_log2m) return false;
  const uint32_t rb = (uint32_t)(pl.fast_k1_wave ? pl.fast_k1_g : pl.g) / 2;   // input bytes per row piece: alignment of every piece
  if (p.payload_off % rb || p.payload_bytes % rb || p.header_bytes % rb || p.frame_bytes % rb ||
      ((uintptr_t)p.frames % 16))
    return false;
  if (pl.fast_k1_wave) {
    // launch-relative 32-bit addressing: frames pointer moved to the frame holding block 0
    if (p.payload_bytes < 2) return false;
    const uint64_t fr0 = p.payload_off / p.payload_bytes;
    const uint64_t rel0 = p.payload_off - fr0 * p.payload_bytes;
    if (rel0 + (uint64_t)nb * pl.block_payload_bytes >= (1ull << 32)) return false;
    KParams q = p;
    q.frames = p.frames + fr0 * p.frame_bytes;
    q.rel0 = (uint32_t)rel0;
    set_fastdiv(q);
    if (h->stg_ready) q.stg = h->stg_cur ? h->stg_cur : h->stg;   // launch_k0_stage has corner-turned this batch
    h->stg_ready = false;
    q.tile_major = p.tile_major = pl.spill_tile_major;   // 2 (R = 2048, paired branches) or 8 (R = 8192) or 0 (K2 of this batch reads what this launch writes)
#ifdef FRBCH_EXPERIMENTS
    if (pl.fast_k1_split && q.stg) {    // persistent over blocks, one 16-wave workgroup per CU
      q.nblk = nb;
      const uint32_t ngrp = (uint32_t)(pl.c2 / 8);
      const uint32_t ny = std::max<uint32_t>(1, std::min<uint32_t>(nb, 256u / std::max<uint32_t>(1, ngrp)));
      hipLaunchKernelGGL(fast::frbch_k1_split, dim3(ngrp, ny), dim3(1024), pl.k1_split_lds, s, q);
      h->kname[KID_K1] = "frbch_k1_split";
      return true;
    }
    if (pl.fast_k1_split) h->kname[KID_K1] = "frbch_k1_wave<3,8,1>";
#endif
    switch (pl.fast_k1_log2m) {
      case 1: launch_k1_wave_t<1>(pl, q, nb, s, h->lane_cus); break;
      case 2: launch_k1_wave_t<2>(pl, q, nb, s, h->lane_cus); break;
      case 3: launch_k1_wave_t<3>(pl, q, nb, s, h->lane_cus); break;
      case 4: launch_k1_wave_t<4>(pl, q, nb, s, h->lane_cus); break;
      case 5: launch_k1_wave_t<5>(pl, q, nb, s, h->lane_cus); break;
      default: return false;
    }
    return true;
  }
  p.tile_major = pl.spill_tile_major == 8 ? 8 : 0;   // (K2 of this batch reads what this launch writes)
  KParams q = p;
  if (h->stg_ready) q.stg = h->stg_cur ? h->stg_cur : h->stg;   // launch_k0_stage has corner-turned this batch
  h->stg_ready = false;
  {   // launch-relative 32-bit addressing where the batch fits (else the kernel divides in 64 bits)
    const uint64_t fr0 = p.payload_off / p.payload_bytes;
    const uint64_t rel0 = p.payload_off - fr0 * p.payload_bytes;
    const uint64_t span = rel0 + (uint64_t)(nb - 1) * pl.block_stride_bytes + pl.block_payload_bytes;
    if (p.payload_bytes >= 2 && span < (1ull << 32)) {
      q.frames = p.frames + fr0 * p.frame_bytes;
      q.rel0 = (uint32_t)rel0;
      q.payload_off = rel0;
      set_fastdiv(q);
    } else {
      q.div_magic = 0;
    }
  }
  switch (pl.fast_k1_log2m) {
    case 1: launch_k1_fast_t<1>(pl, q, nb, s); break;
    case 2: launch_k1_fast_t<2>(pl, q, nb, s); break;
    case 3: launch_k1_fast_t<3>(pl, q, nb, s); break;
    case 4: launch_k1_fast_t<4>(pl, q, nb, s); break;
    case 5: launch_k1_fast_t<5>(pl, q, nb, s); break;
    default: return false;
  }
  return true;
}
bool launch_k2_fast(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  if (pl.fast_k2_lane) {   // 2C = 64 / 128: a whole sequence per lane (pair)
    if (p.tile_major) return false;
    const int pmk = p.pol_mode == 2 ? 2 : (p.pol_mode >= 4 ? 4 : 0);
    const dim3 grid((unsigned)((uint64_t)pl.r * nb * pl.fast_k2_lane / 256));
    const size_t lds = 4 * 64 * (16 * 8 + 16);   // one transposing strip per wave
#define FRBCH_K2L(NHV) do { if (pmk == 2) hipLaunchKernelGGL((fast::frbch_k2_lane<NHV, 2>), grid, dim3(256), lds, s, p); \
                            else if (pmk == 4) hipLaunchKernelGGL((fast::frbch_k2_lane<NHV, 4>), grid, dim3(256), lds, s, p); \
                            else hipLaunchKernelGGL((fast::frbch_k2_lane<NHV, 0>), grid, dim3(256), lds, s, p); } while (0)
    if (pl.fast_k2_lane == 1) FRBCH_K2L(1); else FRBCH_K2L(2);
#undef FRBCH_K2L
    return true;
  }
  if (priv_takes(pl, p, h->priv_grid)) {   // one wave per time sample (kernels_k2priv.inc)
    KParams& k = p;
    k.nblk = nb;
    if (pol_mode_no_sums(k.pol_mode) || (h->cfg.flags & (1u << 20))) k.stat_partial = nullptr;
    const uint64_t ntiles = (uint64_t)nb * (uint64_t)(pl.r / 4);
    const dim3 grid((unsigned)std::min<uint64_t>(ntiles, (uint64_t)h->priv_grid));
    const int pm = k.pol_mode == 2 ? 2 : (k.pol_mode >= 4 ? k.pol_mode : 0);
    const size_t lds = pl.k2_priv_lds;
#define FRBCH_K2P(PMV) do { \
      if (k.out_mode == FRBCH_OUT_CODES) hipLaunchKernelGGL((fast::frbch_k2_priv<PMV, fast::K2P_CODES>), grid, dim3(256), lds, s, k); \
      else if (k.out_mode == FRBCH_OUT_STATS) hipLaunchKernelGGL((fast::frbch_k2_priv<PMV, fast::K2P_STATS>), grid, dim3(256), lds, s, k); \
      else hipLaunchKernelGGL((fast::frbch_k2_priv<PMV, fast::K2P_POWER>), grid, dim3(256), lds, s, k); } while (0)
    if (pm == 2) FRBCH_K2P(2); else if (pm == 4) FRBCH_K2P(4); else if (pm == 5) FRBCH_K2P(5); else FRBCH_K2P(0);
#undef FRBCH_K2P
    return true;
  }
  if (p.out_mode == FRBCH_OUT_STATS) return false;   // (only frbch_k2_priv has a statistics-only form: the engine asks for it nowhere else)
  if (pl.fast_k2_wave) {
    // tscrunch beyond the kernel's tile: rows of its largest tile into the scratch buffer (q), then the sums (p)
    KParams q = p;
    if (pl.k2_two_stage) {
      q.tscr = pl.k2_stage1_tscr;
      q.out_mode = FRBCH_OUT_FLOAT_POWER;
      q.power_out = h->scr2;
      q.row0 = 0;
      q.stat_partial = nullptr;
    }
    KParams& k = pl.k2_two_stage ? q : p;
    switch (pl.fast_k2_log2m) {
      case 0:
        if (!pl.fast_k2_m1) return false;
        launch_k2_wave_t<0>(pl, k, nb, s, h->cfg.flags);
        break;
      case 1: launch_k2_wave_t<1>(pl, k, nb, s, h->cfg.flags); break;
      case 2: launch_k2_wave_t<2>(pl, k, nb, s, h->cfg.flags); break;
      case 3: launch_k2_wave_t<3>(pl, k, nb, s, h->cfg.flags); break;
      case 4: launch_k2_wave_t<4>(pl, k, nb, s, h->cfg.flags); break;
      case 5: launch_k2_wave_t<5>(pl, k, nb, s, h->cfg.flags); break;
      default: return false;
    }
    if (pl.k2_two_stage) {
      p.scr_in = h->scr2;
      p.scr_fact = (uint32_t)pl.k2_two_stage;
      p.scr_rows = (uint64_t)nb * pl.rows_per_block;
      p.stat_partial = nullptr;
      const uint64_t groups = p.scr_rows * (uint64_t)(pl.ncol / 4);
      hipLaunchKernelGGL(fast::frbch_k2_scrunch, dim3((unsigned)std::min<uint64_t>((groups + 255) / 256, 8192)), dim3(256), 0, s, p);
    }
    return true;
  }
  if (pl.k2_two_stage && pl.fast_k2_log2m == 5) {   // barrier K2, four products: two-sample rows into the scratch buffer, then the sums
    KParams q = p;
    q.tscr = pl.k2_stage1_tscr;
    q.out_mode = FRBCH_OUT_FLOAT_POWER;
    q.power_out = h->scr2;
    q.row0 = 0;
    q.stat_partial = nullptr;
    hipLaunchKernelGGL((fast::frbch_k2_fast<5, 1024>), dim3(pl.r / 2, nb), dim3(1024), pl.k2_fast_lds, s, q);
    p.scr_in = h->scr2;
    p.scr_fact = (uint32_t)pl.k2_two_stage;
    p.scr_rows = (uint64_t)nb * pl.rows_per_block;
    p.stat_partial = nullptr;
    const uint64_t groups = p.scr_rows * (uint64_t)(pl.ncol / 4);
    hipLaunchKernelGGL(fast::frbch_k2_scrunch, dim3((unsigned)std::min<uint64_t>((groups + 255) / 256, 8192)), dim3(256), 0, s, p);
    return true;
  }
  switch (pl.fast_k2_log2m) {
#ifdef FRBCH_EXPERIMENTS   // the barrier K2 below 2C = 8192: flags 4 / 8 only
    case 1: launch_k2_fast_t<1>(pl, p, nb, s); break;
    case 2: launch_k2_fast_t<2>(pl, p, nb, s); break;
    case 3: launch_k2_fast_t<3>(pl, p, nb, s); break;
    case 4: launch_k2_fast_t<4>(pl, p, nb, s); break;
#endif
    case 5:
      if (pl.fast_k2_nt == 512) hipLaunchKernelGGL((fast::frbch_k2_fast<5, 512>), dim3(pl.r, nb), dim3(512), pl.k2_fast_lds, s, p);   // one time sample per workgroup (tscrunch 1)
      else hipLaunchKernelGGL((fast::frbch_k2_fast<5, 1024>), dim3(pl.r / std::max(2, pl.tscr), nb), dim3(1024), pl.k2_fast_lds, s, p);
      break;
    default: return false;
  }
  return true;
}
bool launch_k2c_fast(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  if (!pl.coh_fast_c) return false;
  const int tt = 1024 / (16 << pl.coh_fast_c);
  const dim3 grid(pl.r / tt, nb);
  switch (pl.coh_fast_c) {
    case 1: hipLaunchKernelGGL((fast::frbch_k2c_fast<1, 1024>), grid, dim3(1024), pl.k2c_fast_lds, s, p); break;
    case 2: hipLaunchKernelGGL((fast::frbch_k2c_fast<2, 1024>), grid, dim3(1024), pl.k2c_fast_lds, s, p); break;
    case 3: hipLaunchKernelGGL((fast::frbch_k2c_fast<3, 1024>), grid, dim3(1024), pl.k2c_fast_lds, s, p); break;
    case 4: hipLaunchKernelGGL((fast::frbch_k2c_fast<4, 1024>), grid, dim3(1024), pl.k2c_fast_lds, s, p); break;
    case 5: hipLaunchKernelGGL((fast::frbch_k2c_fast<5, 1024>), grid, dim3(1024), pl.k2c_fast_lds, s, p); break;
    default: return false;
  }
  return true;
}
bool launch_k3_fast(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  if (!pl.coh_fast_r || !h->coh_order_m) return false;
  const int np = pl.coh_nt / (16 << pl.coh_fast_r) / 2;
  const dim3 grid(pl.c / np, nb);
  if (pl.coh_nt == 512) {
    if (pl.coh_fast_r != 4) return false;
    if (!(h->cfg.flags & 8u)) {   // wave form: persistent over the (block, channel) tiles, next tile prefetched piecewise
      p.nblk = nb;
      const uint64_t ntiles = (uint64_t)nb * pl.c;
      hipLaunchKernelGGL((fast::frbch_k3_wave<4>), dim3((unsigned)std::min<uint64_t>(ntiles, kK3WaveWgs)), dim3(256), pl.k3_fast_lds, s, p);
      return true;
    }
#ifdef FRBCH_EXPERIMENTS
    hipLaunchKernelGGL((fast::frbch_k3_fast<4, 512>), grid, dim3(512), pl.k3_fast_lds, s, p);
    return true;
#else
    return false;
#endif
  }
  switch (pl.coh_fast_r) {
    case 1: hipLaunchKernelGGL((fast::frbch_k3_fast<1, 1024>), grid, dim3(1024), pl.k3_fast_lds, s, p); break;
    case 2: hipLaunchKernelGGL((fast::frbch_k3_fast<2, 1024>), grid, dim3(1024), pl.k3_fast_lds, s, p); break;
    case 3: hipLaunchKernelGGL((fast::frbch_k3_fast<3, 1024>), grid, dim3(1024), pl.k3_fast_lds, s, p); break;
    case 4: hipLaunchKernelGGL((fast::frbch_k3_fast<4, 1024>), grid, dim3(1024), pl.k3_fast_lds, s, p); break;
    case 5: hipLaunchKernelGGL((fast::frbch_k3_fast<5, 1024>), grid, dim3(1024), pl.k3_fast_lds, s, p); break;
    default: return false;
  }
  return true;
}
template <class K>
int allow_lds(frbch_handle* h, K kern, size_t bytes) {
  CHECK_DEV(h, dev_allow_lds(kern, bytes), "LDS size (fast kernel)");
  return FRBCH_OK;
}
int upload_cf(frbch_handle* h, cf** dst, const std::vector<float>& xy) {
  CHECK_DEV(h, dev_malloc((void**)dst, xy.size() * sizeof(float)), "hipMalloc(fast tables)");
  CHECK_DEV(h, dev_h2d(*dst, xy.data(), xy.size() * sizeof(float), h->stream), "upload fast tables");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  return FRBCH_OK;
}
void fft_tables(int len, std::vector<float>* tw1, std::vector<float>* tw2) {
  const int tps = len / 16, m = len / 256;
  tw1->resize(2 * (size_t)len);
  for (int ka = 0; ka < 16; ++ka)
    for (int q = 0; q < tps; ++q) {
      const double a = -2.0 * M_PI * (double)(((uint64_t)q * ka) % len) / len;
      (*tw1)[2 * ((size_t)ka * tps + q)] = (float)cos(a);
      (*tw1)[2 * ((size_t)ka * tps + q) + 1] = (float)sin(a);
    }
  tw2->resize(2 * (size_t)m * 16);
  for (int kb = 0; kb < m; ++kb)
    for (int c = 0; c < 16; ++c) {
      const double a = -2.0 * M_PI * (double)((c * kb) % tps) / tps;
      (*tw2)[2 * (kb * 16 + c)] = (float)cos(a);
      (*tw2)[2 * (kb * 16 + c) + 1] = (float)sin(a);
    }
}
int setup_fast(frbch_handle* h) {
  const Plan& pl = h->pl;
  int rc;
  std::vector<float> t1, t2;
  if (pl.fast_k1_log2m) {
    fft_tables(pl.r, &t1, &t2);
    if ((rc = upload_cf(h, &h->ftw1_r, t1)) || (rc = upload_cf(h, &h->ftw2_r, t2))) return rc;
    const int tps = pl.r / 16;
    std::vector<float> d1(2 * (size_t)pl.c2 * 16), d2(2 * (size_t)pl.c2 * tps);
    for (int n1 = 0; n1 < pl.c2; ++n1) {
      for (int kc = 0; kc < 16; ++kc) {
        const double a = -2.0 * M_PI * (double)((uint64_t)n1 * kc) / (16.0 * pl.c2);
        d1[2 * ((size_t)n1 * 16 + kc)] = (float)cos(a);
        d1[2 * ((size_t)n1 * 16 + kc) + 1] = (float)sin(a);
      }
      for (int k0 = 0; k0 < tps; ++k0) {
        const double a = -2.0 * M_PI * (double)((uint64_t)n1 * k0) / (double)pl.n;
        d2[2 * ((size_t)n1 * tps + k0)] = (float)cos(a);
        d2[2 * ((size_t)n1 * tps + k0) + 1] = (float)sin(a);
      }
    }
    if ((rc = upload_cf(h, &h->td1, d1)) || (rc = upload_cf(h, &h->td2, d2))) return rc;
#ifdef FRBCH_EXPERIMENTS
    if (pl.fast_k1_split) {
      std::vector<float> h1, h2;
      fft_tables(pl.r / 2, &h1, &h2);
      if ((rc = upload_cf(h, &h->ftw1_h, h1)) || (rc = upload_cf(h, &h->ftw2_h, h2))) return rc;
      if ((rc = allow_lds(h, fast::frbch_k1_split, pl.k1_split_lds))) return rc;
    }
#endif
    if (!h->stg)
      CHECK_DEV(h, dev_malloc((void**)&h->stg, (size_t)pl.maxb * pl.block_payload_bytes), "hipMalloc(staged payload)");
#define FRBCH_AL(L, NWV, WPSV) do { if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<L, NWV, WPSV, false>, pl.k1_fast_lds); \
                                    if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<L, NWV, WPSV, true>, pl.k1_fast_lds); } while (0)
    rc = FRBCH_OK;
    if (pl.fast_k1_wave) switch (pl.fast_k1_log2m) {
      case 1: FRBCH_AL(1, 8, 1); break;
      case 2: FRBCH_AL(2, 8, 1); break;
      case 4:
        FRBCH_AL(4, 8, 2);
        if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<4, 8, 2, false, true>, pl.k1_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<4, 8, 2, true, true>, pl.k1_fast_lds);
        break;
      case 5: FRBCH_AL(5, 8, 4); break;
#ifdef FRBCH_EXPERIMENTS
      default: FRBCH_AL(3, 8, 1); FRBCH_AL(3, 4, 1); FRBCH_AL(3, 8, 2); FRBCH_AL(3, 16, 2); break;
#else
      default: FRBCH_AL(3, 8, 1); break;
#endif
    }
#undef FRBCH_AL
    else switch (pl.fast_k1_log2m) {
      case 1: rc = allow_lds(h, fast::frbch_k1_fast<1>, pl.k1_fast_lds); break;
      case 2: rc = allow_lds(h, fast::frbch_k1_fast<2>, pl.k1_fast_lds); break;
      case 3: rc = allow_lds(h, fast::frbch_k1_fast<3>, pl.k1_fast_lds); break;
      case 4: rc = allow_lds(h, fast::frbch_k1_fast<4>, pl.k1_fast_lds); break;
      default: rc = allow_lds(h, fast::frbch_k1_fast<5>, pl.k1_fast_lds); break;
    }
    if (rc) return rc;
  }
  if (pl.coh_fast_c) {
    fft_tables(pl.c2, &t1, &t2);
    if ((rc = upload_cf(h, &h->ftw1_c, t1)) || (rc = upload_cf(h, &h->ftw2_c, t2))) return rc;
    switch (pl.coh_fast_c) {
      case 1: rc = allow_lds(h, fast::frbch_k2c_fast<1, 1024>, pl.k2c_fast_lds); break;
      case 2: rc = allow_lds(h, fast::frbch_k2c_fast<2, 1024>, pl.k2c_fast_lds); break;
      case 3: rc = allow_lds(h, fast::frbch_k2c_fast<3, 1024>, pl.k2c_fast_lds); break;
      case 4: rc = allow_lds(h, fast::frbch_k2c_fast<4, 1024>, pl.k2c_fast_lds); break;
      default: rc = allow_lds(h, fast::frbch_k2c_fast<5, 1024>, pl.k2c_fast_lds); break;
    }
    if (rc) return rc;
  }
  if (pl.coh_fast_r) {
    switch (pl.coh_fast_r) {
      case 1: rc = allow_lds(h, fast::frbch_k3_fast<1, 1024>, pl.k3_fast_lds); break;
      case 2: rc = allow_lds(h, fast::frbch_k3_fast<2, 1024>, pl.k3_fast_lds); break;
      case 3: rc = allow_lds(h, fast::frbch_k3_fast<3, 1024>, pl.k3_fast_lds); break;
      case 4:
#ifdef FRBCH_EXPERIMENTS
        rc = pl.coh_nt == 512 ? allow_lds(h, fast::frbch_k3_fast<4, 512>, pl.k3_fast_lds) : allow_lds(h, fast::frbch_k3_fast<4, 1024>, pl.k3_fast_lds);
#else
        rc = pl.coh_nt == 512 ? FRBCH_OK : allow_lds(h, fast::frbch_k3_fast<4, 1024>, pl.k3_fast_lds);
#endif
        if (!rc && pl.coh_nt == 512) rc = allow_lds(h, fast::frbch_k3_wave<4>, pl.k3_fast_lds);
        break;
      default: rc = allow_lds(h, fast::frbch_k3_fast<5, 1024>, pl.k3_fast_lds); break;
    }
    if (rc) return rc;
  }
  if (pl.fast_k2_m1) {   // 2C = 256: wave-private K2 only, Kc stays generic
    fft_tables(pl.c2, &t1, &t2);
    if ((rc = upload_cf(h, &h->ftw1_c, t1)) || (rc = upload_cf(h, &h->ftw2_c, t2))) return rc;
    rc = FRBCH_OK;
#define FRBCH_ALLOW0(NWV, PMV) if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<0, NWV, PMV>, pl.k2_fast_lds)
    FRBCH_ALLOW0(2, 0); FRBCH_ALLOW0(2, 2); FRBCH_ALLOW0(2, 4); FRBCH_ALLOW0(4, 0); FRBCH_ALLOW0(4, 2); FRBCH_ALLOW0(4, 4);
    FRBCH_ALLOW0(8, 0); FRBCH_ALLOW0(8, 2); FRBCH_ALLOW0(8, 4);
#undef FRBCH_ALLOW0
    if (rc) return rc;
  }
  if (pl.fast_k2_log2m) {
    fft_tables(pl.c2, &t1, &t2);
    if ((rc = upload_cf(h, &h->ftw1_c, t1)) || (rc = upload_cf(h, &h->ftw2_c, t2))) return rc;
    const size_t kc_lds = ((size_t)pl.c2 + pl.c2 / 8 + 8 + pl.c2) * 8;
    switch (pl.fast_k2_log2m) {
      case 1: rc = allow_lds(h, fast::frbch_kc_fast<1>, kc_lds); break;
      case 2: rc = allow_lds(h, fast::frbch_kc_fast<2>, kc_lds); break;
      case 3: rc = allow_lds(h, fast::frbch_kc_fast<3>, kc_lds); break;
      case 4: rc = allow_lds(h, fast::frbch_kc_fast<4>, kc_lds); break;
      default: rc = allow_lds(h, fast::frbch_kc_fast<5>, kc_lds); break;
    }
    if (rc) return rc;
    const bool big = pl.fast_k2_nt == 1024;
    if (pl.fast_k2_priv) {
      rc = FRBCH_OK;
#define FRBCH_ALLOWP(PMV) do { if (!rc) rc = allow_lds(h, fast::frbch_k2_priv<PMV, fast::K2P_CODES>, pl.k2_priv_lds); \
                               if (!rc) rc = allow_lds(h, fast::frbch_k2_priv<PMV, fast::K2P_POWER>, pl.k2_priv_lds); \
                               if (!rc) rc = allow_lds(h, fast::frbch_k2_priv<PMV, fast::K2P_STATS>, pl.k2_priv_lds); } while (0)
      FRBCH_ALLOWP(0); FRBCH_ALLOWP(2); FRBCH_ALLOWP(4); FRBCH_ALLOWP(5);
#undef FRBCH_ALLOWP
      if (rc) return rc;
    }
    if (pl.fast_k2_wave) {
      rc = FRBCH_OK;
#define FRBCH_ALLOW_(L, NWV, PMV, W) if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<L, NWV, PMV, W>, pl.k2_fast_lds)
#define FRBCH_ALLOW4(L, NWV, PMV, W) FRBCH_ALLOW_(L, NWV, PMV, W)
#define FRBCH_ALLOW(...) FRBCH_ALLOW_SEL(__VA_ARGS__, FRBCH_ALLOW4, FRBCH_ALLOW3)(__VA_ARGS__)
#define FRBCH_ALLOW_SEL(a, b, c, d, NAME, ...) NAME
#define FRBCH_ALLOW3(L, NWV, PMV) FRBCH_ALLOW_(L, NWV, PMV, 1)
#define FRBCH_ALLOW_L(L) FRBCH_ALLOW(L, 2, 0); FRBCH_ALLOW(L, 2, 2); FRBCH_ALLOW(L, 2, 4); FRBCH_ALLOW(L, 4, 0); FRBCH_ALLOW(L, 4, 2); FRBCH_ALLOW(L, 4, 4); FRBCH_ALLOW(L, 8, 0); FRBCH_ALLOW(L, 8, 2); FRBCH_ALLOW(L, 8, 4)
      if (pl.fast_k2_log2m == 5) {
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<5, 8, 2, 4, true>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<5, 8, 0, 4, true>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<5, 8, 2, 4>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<5, 8, 0, 4>, pl.k2_fast_lds);
      } else
      if (pl.fast_k2_log2m == 4) {
        FRBCH_ALLOW(4, 4, 0, 2); FRBCH_ALLOW(4, 4, 2, 2); FRBCH_ALLOW(4, 4, 4, 2);
        FRBCH_ALLOW(4, 8, 0, 2); FRBCH_ALLOW(4, 8, 2, 2); FRBCH_ALLOW(4, 8, 4, 2);
      } else
      if (pl.fast_k2_log2m == 1) { FRBCH_ALLOW_L(1); }
      else if (pl.fast_k2_log2m == 2) { FRBCH_ALLOW_L(2); }
      else {
        if constexpr (kExperiments) { FRBCH_ALLOW_L(3); }
        else { FRBCH_ALLOW(3, 8, 0); FRBCH_ALLOW(3, 8, 2); FRBCH_ALLOW(3, 8, 4); }
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 4, 0, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 4, 2, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 4, 4, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 8, 0, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 8, 2, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 8, 4, 2>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 4, 4, 2, true>, pl.k2_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k2_wave<3, 8, 4, 2, true>, pl.k2_fast_lds);
      }
#undef FRBCH_ALLOW_L
#undef FRBCH_ALLOW
    }
    else switch (pl.fast_k2_log2m) {
#ifdef FRBCH_EXPERIMENTS
      case 1: rc = big ? allow_lds(h, fast::frbch_k2_fast<1, 1024>, pl.k2_fast_lds) : allow_lds(h, fast::frbch_k2_fast<1, 512>, pl.k2_fast_lds); break;
      case 2: rc = big ? allow_lds(h, fast::frbch_k2_fast<2, 1024>, pl.k2_fast_lds) : allow_lds(h, fast::frbch_k2_fast<2, 512>, pl.k2_fast_lds); break;
      case 3: rc = big ? allow_lds(h, fast::frbch_k2_fast<3, 1024>, pl.k2_fast_lds) : allow_lds(h, fast::frbch_k2_fast<3, 512>, pl.k2_fast_lds); break;
      case 4: rc = big ? allow_lds(h, fast::frbch_k2_fast<4, 1024>, pl.k2_fast_lds) : allow_lds(h, fast::frbch_k2_fast<4, 512>, pl.k2_fast_lds); break;
#else
      case 1: case 2: case 3: case 4: rc = fail(h, FRBCH_E_ARG, "the barrier K2 below 8192 branches exists only in experiments builds"); break;
#endif
      default: rc = big ? allow_lds(h, fast::frbch_k2_fast<5, 1024>, pl.k2_fast_lds) : allow_lds(h, fast::frbch_k2_fast<5, 512>, pl.k2_fast_lds); break;
    }
    if (rc) return rc;
  }
  return FRBCH_OK;
}
#else
bool launch_kc_fast(frbch_handle*, KParams&, uint32_t, dev_stream_t) { return false; }
bool launch_k1_fast(frbch_handle*, KParams&, uint32_t, dev_stream_t) { return false; }
bool launch_k2_fast(frbch_handle*, KParams&, uint32_t, dev_stream_t) { return false; }
void launch_k0_stage(frbch_handle*, const KParams&, uint32_t, dev_stream_t) {}
bool launch_k2c_fast(frbch_handle*, KParams&, uint32_t, dev_stream_t) { return false; }
bool launch_k3_fast(frbch_handle*, KParams&, uint32_t, dev_stream_t) { return false; }
int setup_fast(frbch_handle*) { return FRBCH_OK; }
#endif

// dedispersion kernel table in the fine-bin order of the K1 / K3 pair in use (order_m = 0: generic, M: register passes)
int build_chirp(frbch_handle* h, int order_m) {
  const Plan& pl = h->pl;
  ChirpParams cp;
  memset(&cp, 0, sizeof cp);
  cp.chirp = h->chirp;
  cp.c = pl.c; cp.c2 = pl.c2; cp.r = pl.r; cp.log2_r = pl.log2_r;
  cp.usb = h->cfg.bw_mhz > 0 ? 1 : 0;
  cp.order_m = order_m;
  const double abw = fabs(h->cfg.bw_mhz);
  cp.band_edge_mhz = cp.usb ? h->cfg.freq_mhz - abw / 2.0 : h->cfg.freq_mhz + abw / 2.0;
  cp.df_mhz = abw / pl.c;
  cp.dm_over_k = h->cfg.dm / kDmDispersion;
  DEV_LAUNCH(frbch_chirp_build, (pl.n + 255) / 256, 1, 256, 0, h->stream, cp);
  CHECK_DEV(h, dev_check_launch(), "launch chirp build");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  h->coh_order_m = order_m;
  return FRBCH_OK;
}

dev_event_t pool_event(frbch_handle* h);
// K0 + K1 + Kc over nb blocks: frames -> spill, P0.  K0 may run on another stream (`sk`, the back lane's CUs): K1 waits for it.
int launch_front(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s, dev_stream_t sk) {
  const Plan& pl = h->pl;
  p.tile_major = 0;   // set by the K1 that writes that layout
  {
    const bool masked = p.fbad != nullptr;   // blocks that touch invalid / filler frames: the generic K1 zeroes those samples
    if (!masked) {
      launch_k0_stage(h, p, nb, sk);
      if (sk != s && h->stg_ready) {
        const dev_event_t e = pool_event(h);
        dev_event_record(e, sk);
        (void)dev_stream_wait(s, e);
      }
    }
    const double bytes = (double)nb * ((double)pl.block_payload_bytes * p.frame_bytes / p.payload_bytes +
                                       (double)pl.n * 8.0 + (double)pl.c2 * 8.0);
    ProfScope ps(h, s, KID_K1, bytes);
    bool done = false;
    if (masked) {
      if (pl.coherent && h->coh_order_m) {   // from here on the generic K1 / K3 and their bin order
        const int rc = build_chirp(h, 0);
        if (rc) return rc;
      }
    } else if (!pl.coherent) done = launch_k1_fast(h, p, nb, s);
    else if (h->coh_order_m) {
      done = launch_k1_fast(h, p, nb, s);
      if (!done) {   // a start offset the register kernel cannot gather: from here on the generic K1 / K3 and their bin order
        const int rc = build_chirp(h, 0);
        if (rc) return rc;
      }
    }
    if (!done) DEV_LAUNCH(frbch_k1_branch, pl.c2 / pl.g, nb, pl.nthreads, pl.k1_lds, s, p);
  }
  if (!pl.coherent) {
    ProfScope ps(h, s, KID_KC, (double)nb * pl.c2 * 16.0);
    if (!launch_kc_fast(h, p, nb, s)) DEV_LAUNCH(frbch_kc_dcfix, 1, nb, pl.nthreads, pl.kc_lds, s, p);
  }
  CHECK_DEV(h, dev_check_launch(), "launch K1/Kc");
  return FRBCH_OK;
}

int launch_back(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s) {
  const Plan& pl = h->pl;
  const int tile_t = std::max(pl.tt, pl.tscr);
  const double out_b = p.out_mode == FRBCH_OUT_FLOAT_POWER ? (double)pl.ncol * 4.0 : (double)pl.row_bytes;
  const double bytes = (double)nb * ((double)pl.n * 8.0 + (double)pl.rows_per_block * out_b);
#ifndef FRBCH_NO_FAST
  const bool k3_sums = k3_wave_planned(pl, h->cfg.flags) && h->coh_order_m != 0;   // (the generic K3 of a fallen-back launch does not sum)
#else
  const bool k3_sums = false;
#endif
  if ((!((pl.fast_k2_log2m || pl.fast_k2_m1) && pl.fast_k2_wave) || pl.coherent) && !k3_sums) p.stat_partial = nullptr;   // only the wave-private K2 / K3 sum while they write
  if (p.out_mode == FRBCH_OUT_STATS) {   // first pass of the two-pass rescale: the spill is read, nothing but the sums is written
    ProfScope ps(h, s, KID_K2S, (double)nb * (double)pl.n * 8.0);
    if (!launch_k2_fast(h, p, nb, s)) return fail(h, FRBCH_E_STATE, "statistics-only K2 pass without frbch_k2_priv");
    CHECK_DEV(h, dev_check_launch(), "launch K2 (statistics pass)");
    return FRBCH_OK;
  }
  if (pl.coherent) {   // K2c (branches -> channels, x kernel), K3 (back to time, detect), K4 (time-major rows)
    {
      ProfScope ps(h, s, KID_K2, (double)nb * (double)pl.n * 24.0);
      if (!launch_k2c_fast(h, p, nb, s)) DEV_LAUNCH(frbch_k2c_chirp, pl.r / pl.tt, nb, pl.nthreads, pl.k2_lds, s, p);
    }
    {
      ProfScope ps(h, s, KID_K3, (double)nb * ((double)pl.n * 8.0 + (double)pl.rows_per_block * pl.ncol * 4.0));
      if (!launch_k3_fast(h, p, nb, s)) DEV_LAUNCH(frbch_k3_dedisp, pl.c, nb, pl.nthreads, pl.k3_lds, s, p);
    }
    {
      ProfScope ps(h, s, KID_K4, (double)nb * (double)pl.rows_per_block * (pl.ncol * 4.0 + out_b));
      const int tc = pl.ncol < 64 ? (int)pl.ncol : 64;
      const int gx = (int)((pl.rows_per_block + 63) / 64) * (int)(pl.ncol / tc);
#ifndef FRBCH_NO_FAST
      if (pl.ncol % 64 == 0 && pl.rows_per_block % 2 == 0 && pl.c % 4 == 0 && !(h->cfg.flags & 2u))   // (flags & 2: the generic back end)
        hipLaunchKernelGGL(fast::frbch_k4_fast, dim3(gx, nb), dim3(256), 0, s, p);
      else
#endif
      DEV_LAUNCH(frbch_k4_out, gx, nb, pl.nthreads, pl.k4_lds, s, p);
    }
    CHECK_DEV(h, dev_check_launch(), "launch K2c/K3/K4");
    return FRBCH_OK;
  }
  ProfScope ps(h, s, KID_K2, bytes);
  if (!launch_k2_fast(h, p, nb, s)) DEV_LAUNCH(frbch_k2_chan, pl.r / tile_t, nb, pl.nthreads, pl.k2_lds, s, p);
  CHECK_DEV(h, dev_check_launch(), "launch K2");
  return FRBCH_OK;
}

// the table of partial rescale sums (separate statistics pass: partial_chunks rows; sums fused into K2: fused_chunks rows)
int ensure_partial(frbch_handle* h) {
  if (h->partial) return FRBCH_OK;
  const Plan& pl = h->pl;
  h->partial_chunks = 2048;
  h->fused_chunks = 0;
#ifndef FRBCH_NO_FAST
  h->fused_chunks = fused_stat_chunks(pl, h->cfg.flags, h->cfg.pol_mode, h->priv_grid);   // flag bit 20 forces the separate statistics pass
#endif
  const size_t chunks = (size_t)std::max(h->partial_chunks, h->fused_chunks);
  CHECK_DEV(h, dev_malloc((void**)&h->partial, chunks * pl.ncol * 2 * sizeof(double)), "hipMalloc(partials)");
  return FRBCH_OK;
}
// the float rows of a buffered rescale interval (allocated on first use: a scan whose first interval takes the two-pass form never needs it)
int ensure_powbuf(frbch_handle* h) {
  const int rc = ensure_partial(h);
  if (rc || h->powbuf) return rc;
  const Plan& pl = h->pl;
  h->pow_cap_rows = pl.interval_rows + (uint64_t)pl.maxb * pl.rows_per_block;
  CHECK_DEV(h, dev_malloc((void**)&h->powbuf, h->pow_cap_rows * pl.ncol * sizeof(float)), "hipMalloc(power buffer)");
  return FRBCH_OK;
}

// columns per workgroup of frbch_stats_final: eight (whole lines), two when eight would leave fewer than 128 workgroups
static int stat_final_cpw(const Plan& pl) { return pl.ncol < 1024 ? 2 : 8; }

int run_stats(frbch_handle* h, uint64_t rows, dev_stream_t s) {
  const Plan& pl = h->pl;
  StatParams sp;
  memset(&sp, 0, sizeof sp);
  if (h->fused_valid && h->fused_chunks && h->fused_rows == rows) {   // K2 already summed these rows: reduce only
    sp.partial = h->partial;
    sp.rows = rows;
    sp.ncol = (int)pl.ncol;
    sp.c = pl.c;
    sp.nif = pl.nif;
    sp.flip = pl.flip;
    sp.nchunk = h->fused_chunks;
    sp.cpw = stat_final_cpw(pl);
    sp.offset = h->offset;
    sp.scale = h->scale;
    ProfScope ps(h, s, KID_STATS, (double)h->fused_chunks * pl.ncol * 16.0);
    DEV_LAUNCH(frbch_stats_final, (int)((pl.ncol + sp.cpw - 1) / sp.cpw), 1, 256, 256 * 2 * sizeof(double), s, sp);
    CHECK_DEV(h, dev_check_launch(), "launch stats (final)");
    return FRBCH_OK;
  }
  if (!h->powbuf) return fail(h, FRBCH_E_STATE, "rescale statistics: neither fused sums nor buffered rows");
  sp.power = h->powbuf;
  sp.partial = h->partial;
  sp.rows = rows;
  sp.ncol = (int)pl.ncol;
  sp.c = pl.c;
  sp.nif = pl.nif;
  sp.flip = pl.flip;
  // narrow rows (fewer than 64 column groups): the threads of a 64-thread workgroup share the column groups and split the rows
  const int cg = (int)(pl.ncol / 4);
  sp.rsplit = (cg < 64 && 64 % cg == 0) ? 64 / cg : 1;
  sp.nchunk = (int)std::min<uint64_t>((uint64_t)h->partial_chunks / sp.rsplit, std::max<uint64_t>(1, rows / (32 * sp.rsplit)));
  if (sp.nchunk < 1) sp.nchunk = 1;
  sp.rows_per_chunk = (rows + sp.nchunk - 1) / sp.nchunk;
  sp.nchunk = (int)((rows + sp.rows_per_chunk - 1) / sp.rows_per_chunk);
  sp.cpw = stat_final_cpw(pl);
  sp.offset = h->offset;
  sp.scale = h->scale;
  const int gx4 = (int)((pl.ncol / 4 * sp.rsplit + 63) / 64);
  ProfScope ps(h, s, KID_STATS, (double)rows * pl.ncol * 4.0);
  DEV_LAUNCH(frbch_stats_partial, gx4, sp.nchunk, 64, 0, s, sp);
  sp.nchunk *= sp.rsplit;          // rows of partial sums the final reduction adds up (fixed order: deterministic)
  DEV_LAUNCH(frbch_stats_final, (int)((pl.ncol + sp.cpw - 1) / sp.cpw), 1, 256, 256 * 2 * sizeof(double), s, sp);
  CHECK_DEV(h, dev_check_launch(), "launch stats");
  return FRBCH_OK;
}

// Geometry of the lean 8-bit digitiser (frbch_quantise_fast) on `ncu` CUs (0 = the whole chip; negative: |ncu| CUs held by one
// 512-thread workgroup each, see run_quantise): workgroups, threads per workgroup and row phases; false = the generic kernel runs
bool quant_fast_geometry(const frbch_handle* h, int ncu, int wgs_per_cu, uint64_t rp_force, uint64_t* wgs_out, uint64_t* nthr_out, uint64_t* rp_out) {
  const Plan& pl = h->pl;
  if ((h->cfg.flags & kFlagGenericQuant) || h->cfg.nbit_out != 8 || pl.digi_max != 255.0f) return false;
  const uint64_t cg = pl.ncol / 4;
  const bool pow2 = (pl.ncol & (pl.ncol - 1)) == 0 && (pl.c & (pl.c - 1)) == 0;
  if (!pow2 || pl.c < 4 || cg < 64) return false;
  const bool excl = ncu < 0;
  const uint64_t nthr = excl ? 512 : 256;
  uint64_t rp = excl ? (uint64_t)(-ncu) * nthr / cg : (uint64_t)(ncu > 0 ? ncu : 256) * (uint64_t)(wgs_per_cu > 0 ? wgs_per_cu : 3) * 256 / cg;
  if (rp_force) rp = rp_force;
  const uint64_t wgs = rp * cg / nthr;
  const uint64_t pitch = h->out_pitch ? h->out_pitch : (uint64_t)pl.c;
  if (!wgs || wgs * nthr != rp * cg || pitch % 4 || rp * (uint64_t)pl.nif * pitch >= (1ull << 31) || rp * pl.ncol * 4 >= (1ull << 31)) return false;
  *wgs_out = wgs;
  *nthr_out = nthr;
  *rp_out = rp;
  return true;
}

int run_quantise(frbch_handle* h, uint64_t rows, uint8_t* dst, dev_stream_t s, int ncu = 0) {
  const Plan& pl = h->pl;
  QuantParams qp;
  memset(&qp, 0, sizeof qp);
  qp.power = h->powbuf;
  qp.out = dst;
  qp.rows = rows;
  qp.ncol = (int)pl.ncol;
  qp.c = pl.c;
  qp.nif = pl.nif;
  qp.flip = pl.flip;
  qp.nbit = h->cfg.nbit_out;
  qp.offset = h->offset;
  qp.scale = h->scale;
  qp.digi_mean = pl.digi_mean;
  qp.digi_scale = pl.digi_scale;
  qp.digi_max = pl.digi_max;
  const uint64_t total = rows * pl.ncol / 4;
  // grid-stride, 4 groups per thread per trip, up to 32 workgroups per CU of the stream it runs on.  (A bare 16-B-in / 4-B-out
  // stream reads fastest with 8 waves per CU, tools/micro/stream_cus; this kernel carries ~60 VALU instructions per group --
  // index arithmetic, rescale, four digitiser chains -- and needs the waves: 1 / 2 / 4 / 8 / 32 workgroups per CU measured
  // 2.8 / 1.76 / 1.41 / 1.59 / 1.28 ms per 6.4 GB, profiles/r03_overlap_sweep_quantise_lane.txt)
#ifdef FRBCH_EXPERIMENTS
  static const int wgs_env = getenv("FRBCH_QUANT_WGS") ? atoi(getenv("FRBCH_QUANT_WGS")) : 0;
#else
  const int wgs_env = 0;
#endif
  const uint64_t gx = std::min<uint64_t>((total + 256 * 4 - 1) / (256 * 4), (uint64_t)(ncu != 0 ? std::abs(ncu) : 256) * (uint64_t)(wgs_env > 0 ? wgs_env : 32));
  qp.grid_x = (uint32_t)std::max<uint64_t>(1, gx);
  qp.log2_c = 0;
  while ((1 << qp.log2_c) < pl.c) ++qp.log2_c;
  qp.log2_ncol = 0;
  while ((1ull << qp.log2_ncol) < pl.ncol) ++qp.log2_ncol;
  qp.pitch = h->out_pitch ? h->out_pitch : (uint64_t)pl.c;
  ProfScope ps(h, s, KID_QUANT, (double)rows * (pl.ncol * 4.0 + pl.row_bytes));
#ifndef FRBCH_NO_FAST
  {
    // 8-bit codes of power-of-two rows: the lean stream (frbch_quantise_fast): every thread one column group, threads = row
    // phases x column groups.  The loads a thread has in flight are `rphases` rows apart, and the HBM address hash does not like
    // every distance: 3 workgroups per CU (four products at 1024 channels: 192 phases, the loads 3 MiB apart) measured 9.1 ms per
    // 8 IFs of config 3, 1 / 2 / 4 / 6 / 8 / 16 per CU 9.7 / 10.6 / 11.1 / 9.5 / 11.1 / 10.4, odd phase counts 63 / 95 / 127 / 191 /
    // 193 / 255 / 383 / 511: 10.1 / 9.5 / 10.5 / 9.4 / 10.1 / 11.2 / 10.1 / 11.8 (generic kernel: 9.9)
    // ncu < 0: |ncu| workgroups of 512 threads, each reserving more than half the LDS: one per CU, and no wave K1 workgroup (148 KB)
    // beside it -- the digitiser holds |ncu| CUs to itself on a plain stream while the next IF's K1 runs on the others.  (16 loads
    // in flight per thread or 1024 threads per workgroup: the same time; the two kernels together move 5.2 TB/s.)
    const bool excl = ncu < 0;
#ifdef FRBCH_EXPERIMENTS
    static const int rp_env = getenv("FRBCH_QUANT_RP") ? atoi(getenv("FRBCH_QUANT_RP")) : 0;
#else
    const int rp_env = 0;
#endif
    uint64_t wgs = 0, nthr = 0, rp = 0;
    if (quant_fast_geometry(h, ncu, wgs_env, excl ? 0 : (uint64_t)rp_env, &wgs, &nthr, &rp)) {
      qp.grid_x = (uint32_t)wgs;
      qp.rphases = (uint32_t)rp;
      h->kname[KID_QUANT] = "frbch_quantise_fast<8>";
      if (excl) {
        constexpr size_t kHold = 84 * 1024;
        if (!h->quant_lds_allowed) {   // (per handle: the attribute belongs to the handle's device)
          CHECK_DEV(h, dev_allow_lds(fast::frbch_quantise_fast<8, 512>, kHold), "LDS size digitiser");
          h->quant_lds_allowed = true;
        }
        hipLaunchKernelGGL((fast::frbch_quantise_fast<8, 512>), dim3(qp.grid_x), dim3(512), kHold, s, qp);
      } else {
        hipLaunchKernelGGL((fast::frbch_quantise_fast<8, 256>), dim3(qp.grid_x), dim3(256), 0, s, qp);
      }
      CHECK_DEV(h, dev_check_launch(), "launch quantise");
      return FRBCH_OK;
    }
  }
#endif
  DEV_LAUNCH(frbch_quantise, qp.grid_x, 1, 256, 0, s, qp);
  CHECK_DEV(h, dev_check_launch(), "launch quantise");
  return FRBCH_OK;
}

// bytes between the starts of consecutive output rows at d_out, and bytes `rows` rows span from d_out
// (packed rows, or this IF's columns of a wider row buffer: out_pitch values per (row, product) line)
uint64_t out_row_span(const frbch_handle* h) {
  const Plan& pl = h->pl;
  if (!h->out_pitch) return pl.row_bytes;
  const uint64_t bits = pl.row_bytes * 8 / pl.ncol;                 // bits per value
  return h->out_pitch * (uint64_t)pl.nif * bits / 8;
}
uint64_t out_extent(const frbch_handle* h, uint64_t rows) {
  const Plan& pl = h->pl;
  if (!rows) return 0;
  if (!h->out_pitch) return rows * pl.row_bytes;
  const uint64_t bits = pl.row_bytes * 8 / pl.ncol;                 // bits per value
  const uint64_t line = h->out_pitch * bits / 8, seg = (uint64_t)pl.c * bits / 8;
  return (rows * pl.nif - 1) * line + seg;
}

// Close the rescale interval that sits at the front of powbuf: statistics over `stat_rows` rows,
// then digitise `emit_rows` rows into dst and keep the rest for the next interval.
struct Chain;
int chain_quant_lane_cus(const Chain* c);
dev_stream_t chain_quant_stream(Chain* c, dev_stream_t s, int* ncu);
void chain_quant_done(Chain* c, dev_stream_t sq);
int finalize_interval(frbch_handle* h, uint64_t stat_rows, uint8_t* d_out, size_t cap, uint64_t* rows_written,
                      dev_stream_t s0, Chain* ch = nullptr) {
  const Plan& pl = h->pl;
  if (h->quant_busy) {   // a digitiser of the previous interval (and its move of the remaining rows) still works on the power buffer
    (void)dev_stream_wait(s0, h->quant_ev);
    h->quant_busy = false;
  }
  int rc = run_stats(h, stat_rows, s0);
  if (rc) return rc;
  // the back lane when the digitiser can run beside the next K1 (on CUs it holds by an LDS reservation: only the lean kernel does that)
  dev_stream_t s = s0;
  {
    const int lane = chain_quant_lane_cus(ch);
    uint64_t a_ = 0, b_ = 0, c_ = 0;
    if (lane > 0 || (lane < 0 && quant_fast_geometry(h, lane, 0, 0, &a_, &b_, &c_))) s = chain_quant_stream(ch, s0, &h->quant_lane_cus);
  }
  h->have_scale = true;
  if (h->cfg.rescale_constant) h->scale_frozen = true;
  const uint64_t emit_rows = h->scale_frozen ? h->pow_rows : stat_rows;
  if (out_extent(h, *rows_written + emit_rows) > cap)
    return fail(h, FRBCH_E_CAPACITY, "output buffer too small for the rows of a completed rescale interval");
  rc = run_quantise(h, emit_rows, d_out + *rows_written * out_row_span(h), s, s != s0 ? h->quant_lane_cus : 0);
  if (rc) return rc;
  *rows_written += emit_rows;
  h->rows_out += emit_rows;
  const uint64_t rest = h->pow_rows - emit_rows;
  // forward chunked move (chunks no longer than the shift distance never overlap)
  uint64_t done = 0;
  while (done < rest) {
    const uint64_t len = std::min<uint64_t>(emit_rows, rest - done);
    CHECK_DEV(h, dev_d2d(h->powbuf + done * pl.ncol, h->powbuf + (emit_rows + done) * pl.ncol,
                         len * pl.ncol * sizeof(float), s), "move power rows");
    done += len;
  }
  h->pow_rows = rest;
  h->fused_rows = 0;
  h->fused_valid = false;   // (re-armed when an interval starts on an empty buffer)
  if (s != s0) {            // later work of this handle on the power buffer / the codes is ordered behind the digitiser
    if (!h->quant_ev_made) { (void)dev_event_create_sync(&h->quant_ev); h->quant_ev_made = true; }
    dev_event_record(h->quant_ev, s);
    h->quant_busy = true;
    chain_quant_done(ch, s);
  }
  return FRBCH_OK;
}

bool fused_ok(const frbch_handle* h) { return h->scale_frozen; }

// ---- two-pass rescale of a first interval (`-c`, which the reference always passes, process_vdif.py:157,160) -------------------
// The buffered form writes the float rows of the interval (8 B per dual-pol sample with four products), reads them back in the
// digitiser and moves 26 B per sample behind K1; with frbch_k2_priv the batch's spill is still resident when the interval is
// complete, so K2 runs over it twice instead: a statistics-only pass (sums, no rows), frbch_stats_final, then the digitising pass --
// 18 B per sample, and the SAME float arithmetic in both passes, so the codes are what the buffered form produces from the same
// offset / scale.  Taken when the interval starts with this batch and either ends inside it or the batch is the last of the call
// (the flush, or the next call, then finds the batch deferred).  MEASURED SLOWER than the buffered form (round 4):
// both K2 passes are bound by their waves' instruction chains (~1.5 ms each per IF of config 3), while the buffered form's extra 16 B
// per sample stream at 5.2 - 5.45 TB/s partly beside the next IF's K1.  Opt-in by flag bit 27; the default stays buffered.
bool twopass_usable(const frbch_handle* h) {
#ifndef FRBCH_NO_FAST
  const Plan& pl = h->pl;
  return pl.fast_k2_priv && h->priv_grid > 0 && h->cfg.rescale_constant && pl.interval_rows > 0 && !pl.k2_two_stage &&
         (h->cfg.flags & (1u << 27)) && fused_stat_chunks(pl, h->cfg.flags, h->cfg.pol_mode, h->priv_grid) > 0;
#else
  return false;
#endif
}
// second pass over a deferred batch: offset / scale are final, K2 digitises the batch's rows into d_out
int deferred_emit(frbch_handle* h, uint8_t* d_out, size_t cap, uint64_t* rows_written, dev_stream_t s) {
  frbch_handle::Deferred& d = h->deferred;
  if (out_extent(h, *rows_written + d.rows) > cap) return fail(h, FRBCH_E_CAPACITY, "output buffer too small for the rows of a completed rescale interval");
  KParams p = d.p;
  p.offset = h->offset;
  p.scale = h->scale;
  p.out_mode = FRBCH_OUT_CODES;
  p.code_out = d_out;
  p.row0 = *rows_written;
  p.out_pitch = h->out_pitch ? h->out_pitch : (uint64_t)h->pl.c;
  p.stat_partial = nullptr;
  const int rc = launch_back(h, p, d.nb, s);
  if (rc) return rc;
  *rows_written += d.rows;
  h->rows_out += d.rows;
  d.active = false;
  return FRBCH_OK;
}
// more blocks arrive while a batch is deferred and its interval is still open: the batch becomes the front of the buffered
// interval after all (one K2 pass writes its float rows; its sums are already in the partial table)
int deferred_materialise(frbch_handle* h, dev_stream_t s) {
  frbch_handle::Deferred& d = h->deferred;
  int rc = ensure_powbuf(h);
  if (rc) return rc;
  if (d.rows > h->pow_cap_rows) return fail(h, FRBCH_E_STATE, "power buffer overflow");
  KParams p = d.p;
  p.out_mode = FRBCH_OUT_FLOAT_POWER;
  p.power_out = h->powbuf;
  p.row0 = 0;
  p.stat_partial = nullptr;
  rc = launch_back(h, p, d.nb, s);
  if (rc) return rc;
  h->pow_rows = d.rows;
  d.active = false;
  return FRBCH_OK;
}

// =============================================================================================
// Two lanes.  The front half of a batch (K0, K1, Kc) is bound by the instruction chain of its
// waves and leaves HBM more than half idle; the back half (K2, statistics, digitiser) is bound by HBM and leaves the
// vector units idle.  They run on two streams whose CU masks split the chip, so that the front of batch b + 1 overlaps the
// back of batch b.  Mask bit i is CU i / 8 of XCD i % 8 (the driver deals the bits round-robin over the XCDs): a lane of
// the first 8 k bits owns k CUs of every XCD, and the workgroup -> XCD round-robin the kernels' tile orders rely on holds
// inside a lane as on the whole chip.  Placement affects speed only: every dependency is a stream-ordered event.
// =============================================================================================
struct Lanes {
  dev_stream_t f = 0, b = 0, b2 = 0;   // front lane; back lane; a second stream on the back lane's CUs (K0 beside K2)
  int ncu = 0, ncu_f = 0;
  bool ok = false;
};
std::mutex g_lanes_mutex;
std::vector<std::pair<std::pair<int, int>, Lanes*>> g_lanes;   // (device, front CUs) -> lanes; live until the process ends

Lanes* get_lanes(int device, int ncu_front, bool plain = false) {
  std::lock_guard<std::mutex> lk(g_lanes_mutex);
  const int key = plain ? -ncu_front : ncu_front;
  for (auto& e : g_lanes)
    if (e.first.first == device && e.first.second == key) return e.second->ok ? e.second : nullptr;
  Lanes* ln = new Lanes();
  g_lanes.push_back({{device, key}, ln});
  ln->ncu = dev_cu_count(device);
  // (a failed creation leaves the entry with ok = false: the streams made so far are destroyed, later calls get nullptr)
  auto three_plain = [&]() {
    if (dev_stream_create(&ln->f) == 0 && dev_stream_create(&ln->b) == 0 && dev_stream_create(&ln->b2) == 0) return true;
    if (ln->f) dev_stream_destroy(ln->f);
    if (ln->b) dev_stream_destroy(ln->b);
    if (ln->b2) dev_stream_destroy(ln->b2);
    ln->f = ln->b = ln->b2 = 0;
    (void)dev_last_error_string();
    return false;
  };
  if (plain && ln->ncu > 0) {
    // mode 3: plain streams (no CU masks: a masked queue costs every launch of the process 50 - 100 us while it is active,
    // profiles/NOTES.md); the digitiser claims its CUs by its LDS reservation, ncu_f is what is left for the K1 beside it
    if (ncu_front < 8 || ncu_front > ln->ncu - 8) return nullptr;
    if (!three_plain()) return nullptr;
    ln->ncu_f = ncu_front;
    ln->ok = true;
    return ln;
  }
  if (ncu_front >= ln->ncu && ln->ncu > 0) {
    // no partition: plain streams.  Kernels of the two lanes share every CU as far as its registers, LDS and wave slots go
    // (the digitiser's 4-wave workgroups fit beside the wave K1's eight 216-register waves: one per CU)
    if (!three_plain()) return nullptr;
    ln->ncu_f = ln->ncu;
    ln->ok = true;
    return ln;
  }
#ifndef FRBCH_EXPERIMENTS
  return nullptr;   // CU-masked lanes (overlap modes 1 and 2) were measured slower: experiments builds only
#else
  if (ln->ncu < 32 || ln->ncu % 8 || ncu_front < 8 || ncu_front > ln->ncu - 8 || ncu_front % 8) return nullptr;
  const uint32_t words = (uint32_t)((ln->ncu + 31) / 32);
  std::vector<uint32_t> mf(words, 0u), mb(words, 0u);
  for (int i = 0; i < ln->ncu; ++i) (i < ncu_front ? mf : mb)[(size_t)i >> 5] |= 1u << (i & 31);
  if (dev_stream_create_masked(&ln->f, mf.data(), words) != 0 || dev_stream_create_masked(&ln->b, mb.data(), words) != 0 ||
      dev_stream_create_masked(&ln->b2, mb.data(), words) != 0) {
    if (ln->f) dev_stream_destroy(ln->f);
    if (ln->b) dev_stream_destroy(ln->b);
    if (ln->b2) dev_stream_destroy(ln->b2);
    ln->f = ln->b = ln->b2 = 0;
    (void)dev_last_error_string();
    return nullptr;
  }
  ln->ncu_f = ncu_front;
  ln->ok = true;
  return ln;
#endif
}

dev_event_t pool_event(frbch_handle* h) {
  if (h->evpool.empty()) {
    h->evpool.resize(64);
    for (auto& e : h->evpool) (void)dev_event_create_sync(&e);
  }
  const dev_event_t e = h->evpool[h->evnext];
  h->evnext = (h->evnext + 1) % h->evpool.size();
  return e;
}

// The stages of one API call (or of one scan call, across its IFs) in the order they are queued.
struct Chain {
  Lanes* ln = nullptr;            // null: every stage on `user`, as queued
  dev_stream_t user = 0;
  frbch_handle* owner = nullptr;  // whose event pool is used
  uint32_t stages_total = 0;      // (front, back) pairs the chain will run; the last back stage runs on the whole chip
  uint32_t fronts = 0, backs = 0;
  bool k0_back = false;           // K0 beside the back lane's kernels instead of in front of K1 on the front lane
  int mode = 1;                   // 1: K2 / statistics / digitiser on the back lane, K0 / K1 / Kc on the front lane
                                  // 2: only the digitiser of a completed interval on the (small) back lane, beside the NEXT
                                  //    front stage on the front lane; K2 keeps the whole chip
  dev_event_t ev_entry{}, ev_front{}, ev_back{}, ev_q{};
  dev_stream_t s_front = 0, s_back = 0;   // streams of the last front / back stage queued
  bool f_rooted = false, b_rooted = false, b2_rooted = false;
  bool q_pending = false, have_q = false; // mode 2: a digitiser runs on the back lane (the next front stage goes beside it)
  bool front_beside_q = false;            // mode 3: the front stage being queued shares the chip with a digitiser
};

void chain_begin(Chain* c, frbch_handle* owner, dev_stream_t user, Lanes* ln, uint32_t stages_total, bool k0_back, int mode) {
  *c = Chain();
  c->owner = owner;
  c->user = user;
  c->ln = stages_total >= 2 ? ln : nullptr;
  c->stages_total = stages_total;
  c->mode = mode;
  c->k0_back = k0_back && mode == 1;
  if (c->ln) {
    c->ev_entry = pool_event(owner);
    dev_event_record(c->ev_entry, user);
  }
}
// stream of the next front stage: the first one has the chip to itself (nothing to overlap with yet)
dev_stream_t chain_front_stream(Chain* c) {
  if (!c->ln || c->fronts == 0) return c->user;
  if (c->mode == 3) {   // the caller's stream throughout; only the K1 grid changes while a digitiser holds part of the chip
    c->front_beside_q = c->q_pending;
    c->q_pending = false;
    return c->user;
  }
  if (c->mode == 2) {
    if (!c->q_pending) return c->user;
    c->q_pending = false;
    if (!c->f_rooted) {
      (void)dev_stream_wait(c->ln->f, c->ev_entry);
      c->f_rooted = true;
    }
    if (c->backs) (void)dev_stream_wait(c->ln->f, c->ev_back);   // the K2 before it had the whole chip
    return c->ln->f;
  }
  if (!c->f_rooted) {
    (void)dev_stream_wait(c->ln->f, c->ev_entry);
    if (c->s_front == c->user && c->fronts) (void)dev_stream_wait(c->ln->f, c->ev_front);   // behind the whole-chip first front
    c->f_rooted = true;
  }
  return c->ln->f;
}
dev_stream_t chain_k0_stream(Chain* c, dev_stream_t front) {
  if (!c->ln || !c->k0_back || front == c->user) return front;
  if (!c->b2_rooted) {
    (void)dev_stream_wait(c->ln->b2, c->ev_entry);
    c->b2_rooted = true;
  }
  return c->ln->b2;
}
void chain_front_done(Chain* c, dev_stream_t sf) {
  c->s_front = sf;
  c->fronts++;
  if (c->ln) {
    c->ev_front = pool_event(c->owner);
    dev_event_record(c->ev_front, sf);
  }
}
// stream of the next back stage (ordered behind its front stage and the previous back stage)
dev_stream_t chain_back_stream(Chain* c) {
  if (!c->ln) return c->user;
  if (c->mode >= 2) {
    if (c->s_front != c->user) (void)dev_stream_wait(c->user, c->ev_front);
    return c->user;
  }
  const bool last = c->backs + 1 >= c->stages_total;
  const dev_stream_t sb = last ? c->user : c->ln->b;
  if (sb == c->ln->b && !c->b_rooted) {
    (void)dev_stream_wait(sb, c->ev_entry);
    c->b_rooted = true;
  }
  if (c->s_front != sb) (void)dev_stream_wait(sb, c->ev_front);
  if (c->backs && c->s_back != sb) (void)dev_stream_wait(sb, c->ev_back);
  return sb;
}
void chain_back_done(Chain* c, dev_stream_t sb) {
  c->s_back = sb;
  c->backs++;
  if (c->ln) {
    c->ev_back = pool_event(c->owner);
    dev_event_record(c->ev_back, sb);
  }
}
// everything the chain queued is ordered in front of what follows on the caller's stream
// mode 2: the stream the digitiser of a completed interval goes to -- the back lane while another front stage is still to
// come (it runs beside that stage's K1), else the stream `s` of the statistics in front of it
int chain_quant_lane_cus(const Chain* c) {   // CUs the digitiser would get on the back lane (negative: held by its LDS reservation); 0 = no lane
  if (!c || !c->ln || c->mode < 2 || c->fronts >= c->stages_total) return 0;
  const int n = c->ln->ncu_f >= c->ln->ncu ? c->ln->ncu : c->ln->ncu - c->ln->ncu_f;
  return c->mode == 3 ? -n : n;
}
dev_stream_t chain_quant_stream(Chain* c, dev_stream_t s, int* ncu) {
  if (!chain_quant_lane_cus(c)) return s;
  const dev_stream_t sq = c->ln->b;
  *ncu = chain_quant_lane_cus(c);
  if (!c->b_rooted) {
    (void)dev_stream_wait(sq, c->ev_entry);
    c->b_rooted = true;
  }
  const dev_event_t e = pool_event(c->owner);
  dev_event_record(e, s);               // behind the statistics
  (void)dev_stream_wait(sq, e);
  return sq;
}
void chain_quant_done(Chain* c, dev_stream_t sq) {
  if (!c || !c->ln || c->mode < 2 || sq != c->ln->b) return;
  c->ev_q = pool_event(c->owner);
  dev_event_record(c->ev_q, sq);
  c->q_pending = true;
  c->have_q = true;
}
void chain_end(Chain* c) {
  if (!c->ln) return;
  if (c->have_q) (void)dev_stream_wait(c->user, c->ev_q);
  if (c->fronts && c->s_front != c->user) (void)dev_stream_wait(c->user, c->ev_front);
  if (c->backs && c->s_back != c->user) (void)dev_stream_wait(c->user, c->ev_back);
  // (every K0 on the second back-lane stream was waited for by its K1, whose back stage is ordered above)
}
// more work was queued on the stream of the last back stage (the digitiser of a flushed interval): later stages and the
// end of the chain are ordered behind it
void chain_back_touch(Chain* c) {
  if (!c->ln || !c->backs) return;
  c->ev_back = pool_event(c->owner);
  dev_event_record(c->ev_back, c->s_back);
}

// front CUs / batches per call requested through frbch_config::overlap (0 = automatic)
int overlap_front_cus(const frbch_handle* h) {
  const uint32_t v = h->cfg.overlap & 0xFFFFu;
#ifdef FRBCH_EXPERIMENTS
  static const int env = getenv("FRBCH_FRONT_CUS") ? atoi(getenv("FRBCH_FRONT_CUS")) : -1;
  if (env >= 0) return env;
#endif
  if (v == 1) return 0;                       // overlap off
  if (v) return (int)(v / 8 * 8);
  // automatic.  K1 against K2 (mode 1): never -- both scale with their share of the CUs (K2 on 96 CUs takes 2.4x its
  // whole-chip time), splitting the chip between them only adds launches (profiles/r03_overlap_sweep_*.txt).
  // The digitiser of a completed interval beside the next IF's K1 (mode 3, a scan): yes when it is a stream of four products
  // -- 6.4 GB per IF, HBM-bound on the whole chip with most CUs idle, while K1 is bound by its waves and leaves HBM half idle.
  // 80 of 256 CUs for the digitiser make both take ~2.3 ms (1.55 + 1.14 one after the other): 39.5 -> 36.4 ms per 8-IF step of
  // config 3, the two together moving 5.2 TB/s.  One product (1.6 GB): what it hides is what K1 loses on fewer CUs: off.
  const Plan& pl = h->pl;
  if (pl.nif == 4 && h->cfg.nbit_out == 8 && pl.fast_k1_log2m == 3 && pl.fast_k1_wave && !pl.fast_k1_split && h->lane_ncu >= 64)
    return h->lane_ncu * 11 / 16 / 8 * 8;
  return 0;
}
int overlap_mode(const frbch_handle* h) {
  const uint32_t m = (h->cfg.overlap >> 24) & 3u;
#ifdef FRBCH_EXPERIMENTS
  static const int env = getenv("FRBCH_OVERLAP_MODE") ? atoi(getenv("FRBCH_OVERLAP_MODE")) : 0;
  if (env > 0) return env;
#endif
  return m ? (int)m : 3;
}
bool overlap_usable(const frbch_handle* h) {
  const Plan& pl = h->pl;
  return !pl.coherent && pl.fast_k1_log2m && pl.fast_k1_wave && (pl.fast_k2_log2m || pl.fast_k2_m1) && pl.fast_k2_wave &&
         overlap_front_cus(h) >= 8;
}
uint32_t overlap_batches(const frbch_handle* h, uint64_t nblocks) {
  uint32_t v = (h->cfg.overlap >> 16) & 0xFFu;
#ifdef FRBCH_EXPERIMENTS
  static const int env = getenv("FRBCH_PIPE_BATCHES") ? atoi(getenv("FRBCH_PIPE_BATCHES")) : 0;
  if (env > 0) v = (uint32_t)env;
#endif
  if (!v) v = (uint32_t)std::min<uint64_t>(4, nblocks / 24);    // batches of at least 24 blocks (ramp-up and tail of the persistent kernels)
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(v, nblocks));
}

// Transform `nblocks` blocks starting `payload_off` bytes into the payload stream of d_frames.
int engine_feed_run(frbch_handle* h, const uint8_t* d_frames, uint32_t frame_bytes, uint32_t header_bytes,
                    uint64_t payload_off, uint64_t nblocks, uint8_t* d_out, size_t cap, uint64_t* rows_written,
                    dev_stream_t s, const uint32_t* d_fbad, Chain* chain);

// `h_bad` (optional): one flag per frame of d_frames, 1 = the frame is flagged invalid or is a filler for a missing frame
// number; `d_fbad` the same as a bitmap on the device.  Blocks that touch such a frame go through the generic K1, which
// reads their samples as 0; all other blocks take the fast kernels as before.
// `chain` (optional): the caller's chain of stages (a scan queues several IFs into one); else the call runs its own.
int engine_feed(frbch_handle* h, const uint8_t* d_frames, uint32_t frame_bytes, uint32_t header_bytes,
                uint64_t payload_off, uint64_t nblocks, uint8_t* d_out, size_t cap, uint64_t* rows_written,
                dev_stream_t s, const uint8_t* h_bad = nullptr, uint64_t nfr_bad = 0, const uint32_t* d_fbad = nullptr,
                Chain* chain = nullptr) {
  const Plan& pl = h->pl;
  *rows_written = 0;
  bool any = false;
  for (uint64_t f = 0; h_bad && f < nfr_bad && !any; ++f) any = h_bad[f] != 0;
  if (!any) return engine_feed_run(h, d_frames, frame_bytes, header_bytes, payload_off, nblocks, d_out, cap, rows_written, s, nullptr, chain);
  const uint64_t pb = frame_bytes - header_bytes;
  auto dirty = [&](uint64_t b) {
    const uint64_t a0 = payload_off + b * pl.block_stride_bytes, a1 = a0 + pl.block_payload_bytes - 1;
    for (uint64_t f = a0 / pb; f <= a1 / pb && f < nfr_bad; ++f)
      if (h_bad[f]) return true;
    return false;
  };
  for (uint64_t b0 = 0; b0 < nblocks;) {
    const bool d0 = dirty(b0);
    uint64_t b1 = b0 + 1;
    while (b1 < nblocks && dirty(b1) == d0) ++b1;
    const int rc = engine_feed_run(h, d_frames, frame_bytes, header_bytes, payload_off + b0 * pl.block_stride_bytes, b1 - b0, d_out, cap,
                                   rows_written, s, d0 ? d_fbad : nullptr, nullptr);
    if (rc) return rc;
    b0 = b1;
  }
  return FRBCH_OK;
}

// batches a run of `nblocks` blocks is cut into: equal sizes (no short tail launch), at most maxb blocks each, and -- when
// the stages overlap -- at most half the spill (two regions in flight) and at least `want` batches
static void plan_batches(const Plan& pl, uint64_t nblocks, bool overlap, uint32_t want, uint64_t* nbatch, uint64_t* per) {
  const uint64_t cap = overlap ? std::max<uint64_t>(1, pl.maxb / 2) : pl.maxb;
  uint64_t nb = (nblocks + cap - 1) / cap;
  if (overlap) nb = std::max<uint64_t>(nb, want);
  nb = std::max<uint64_t>(1, std::min<uint64_t>(nb, nblocks));
  *nbatch = nb;
  *per = nblocks ? (nblocks + nb - 1) / nb : 0;
  if (*per) *nbatch = (nblocks + *per - 1) / *per;
}
uint64_t feed_stage_count(const frbch_handle* h, uint64_t nblocks, bool overlap) {
  uint64_t nbatch = 0, per = 0;
  plan_batches(h->pl, nblocks, overlap, overlap ? overlap_batches(h, nblocks) : 1, &nbatch, &per);
  return nblocks ? nbatch : 0;
}

// one run of blocks; *rows_written is the running row count of the call (rows land behind those already written)
int engine_feed_run(frbch_handle* h, const uint8_t* d_frames, uint32_t frame_bytes, uint32_t header_bytes,
                    uint64_t payload_off, uint64_t nblocks, uint8_t* d_out, size_t cap, uint64_t* rows_written,
                    dev_stream_t s, const uint32_t* d_fbad, Chain* outer) {
  const Plan& pl = h->pl;
  if (!nblocks) return FRBCH_OK;
  // own chain unless the caller brought one; blocks that touch flagged frames run in queue order on the caller's stream
  Chain own;
  Chain* ch = outer;
  const bool may_overlap = !d_fbad && overlap_usable(h);
  Lanes* ln = nullptr;
  if (!ch && may_overlap) ln = get_lanes(h->device, overlap_front_cus(h), overlap_mode(h) == 3);
  if (!ch && overlap_mode(h) != 1) ln = nullptr;     // (mode 2 only overlaps across the IFs of a scan: the caller's chain)
  const bool overlap = ch ? (ch->ln != nullptr && ch->mode == 1) : (ln != nullptr);   // batches cut for two regions in flight
  uint64_t nbatch = 0, per = 0;
  plan_batches(pl, nblocks, overlap, overlap ? overlap_batches(h, nblocks) : 1, &nbatch, &per);
  if (!ch) {
    chain_begin(&own, h, s, nbatch >= 2 ? ln : nullptr, (uint32_t)nbatch, pl.nif < 4, 1);
    ch = &own;
  }
  const uint32_t nreg = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(1, pl.maxb / per));
  if (!h->region_ev_made) {
    for (auto& e : h->region_ev) (void)dev_event_create_sync(&e);
    h->region_ev_made = true;
  }
  const uint64_t spill_blk = (uint64_t)(pl.c2 / pl.g) * pl.gs;      // cf per block of the spill
  int rc = FRBCH_OK;
  // a batch deferred by the two-pass rescale whose interval goes on: its float rows are written NOW, before the front stage of
  // the next batch overwrites the spill they come from
  if (h->deferred.active && !fused_ok(h)) rc = deferred_materialise(h, ch->user);
  for (uint64_t b0 = 0; b0 < nblocks && !rc; b0 += per) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(per, nblocks - b0);
    const uint32_t reg = (ch->ln && ch->mode == 1) ? (h->next_region++ % nreg) : 0;
    const uint64_t rb0 = (uint64_t)reg * per;                        // first block of the region inside the work buffers
    KParams p = base_params(h);
    p.spill += rb0 * spill_blk;
    p.s_dc += rb0 * pl.c2;
    p.p0 += rb0 * pl.c2;
    h->stg_cur = h->stg ? h->stg + rb0 * pl.block_payload_bytes : nullptr;
    p.fbad = d_fbad;
    p.fbad_frame0 = 0;
    p.frames = d_frames;
    p.frame_bytes = frame_bytes;
    p.header_bytes = header_bytes;
    p.payload_bytes = frame_bytes - header_bytes;
    p.payload_off = payload_off + b0 * pl.block_stride_bytes;
    // ---- front: K0, K1, Kc --------------------------------------------------------------------------------------
    const dev_stream_t sf = chain_front_stream(ch);
    const bool region_wait = h->region_busy[reg] && (ch->ln || outer);   // the back stage that read this region last must be through
    if (region_wait) (void)dev_stream_wait(sf, h->region_ev[reg]);
    h->region_busy[reg] = false;
    const dev_stream_t sk = chain_k0_stream(ch, sf);
    if (sk != sf && region_wait) (void)dev_stream_wait(sk, h->region_ev[reg]);
    h->lane_cus = (ch->ln && (sf == ch->ln->f || (ch->mode == 3 && ch->front_beside_q))) ? ch->ln->ncu_f : 0;
    rc = launch_front(h, p, nb, sf, sk);
    h->lane_cus = 0;
    if (rc) break;
    chain_front_done(ch, sf);
    // ---- back: K2 (+ statistics and digitiser of a completed interval) ------------------------------------------
    const dev_stream_t sb = chain_back_stream(ch);
    if (h->quant_busy) {   // a digitiser of this handle on the back lane still reads the power buffer
      (void)dev_stream_wait(sb, h->quant_ev);
      h->quant_busy = false;
    }
    const uint64_t rows = (uint64_t)nb * pl.rows_per_block;
    if (fused_ok(h)) {
      if (out_extent(h, *rows_written + rows) > cap) { rc = fail(h, FRBCH_E_CAPACITY, "output buffer too small"); break; }
      p.out_mode = FRBCH_OUT_CODES;
      p.code_out = d_out;
      p.row0 = *rows_written;
      rc = launch_back(h, p, nb, sb);
      if (rc) break;
      *rows_written += rows;
      h->rows_out += rows;
    } else if (!h->deferred.active && h->pow_rows == 0 && twopass_usable(h) && p.tile_major == 2 &&
               (rows >= pl.interval_rows || b0 + per >= nblocks)) {
      // ---- two-pass rescale: statistics-only pass now, the digitising pass once the interval is complete ----------------------
      rc = ensure_partial(h);
      if (rc) break;
      if (dev_memset(h->partial, 0, (size_t)h->fused_chunks * pl.ncol * 2 * sizeof(double), sb) != 0) { rc = fail(h, FRBCH_E_DEVICE, "clear partial sums"); break; }
      p.out_mode = FRBCH_OUT_STATS;
      p.row0 = 0;
      p.stat_partial = h->partial;
      p.stat_limit = pl.interval_rows;
      rc = launch_back(h, p, nb, sb);
      if (rc) break;
      h->fused_rows = std::min<uint64_t>(rows, pl.interval_rows);
      h->fused_valid = true;
      h->deferred.active = true;
      h->deferred.p = p;
      h->deferred.nb = nb;
      h->deferred.rows = rows;
      if (rows >= pl.interval_rows) {   // the interval ends inside this batch: offset / scale now, then every row of the batch
        rc = run_stats(h, pl.interval_rows, sb);
        if (rc) break;
        h->have_scale = true;
        h->scale_frozen = true;        // (-c: twopass_usable)
        h->fused_rows = 0;
        h->fused_valid = false;
        rc = deferred_emit(h, d_out, cap, rows_written, sb);
        if (rc) break;
      }
    } else {
      rc = ensure_powbuf(h);
      if (rc) break;
      if (h->pow_rows + rows > h->pow_cap_rows) { rc = fail(h, FRBCH_E_STATE, "power buffer overflow"); break; }
      p.out_mode = FRBCH_OUT_FLOAT_POWER;
      p.power_out = h->powbuf;
      p.row0 = h->pow_rows;
      if (h->pow_rows == 0 && h->fused_chunks) {   // an interval starts here: K2 can sum it while writing it
        if (dev_memset(h->partial, 0, (size_t)h->fused_chunks * pl.ncol * 2 * sizeof(double), sb) != 0) { rc = fail(h, FRBCH_E_DEVICE, "clear partial sums"); break; }
        h->fused_rows = 0;
        h->fused_valid = true;
      }
      if (h->fused_valid) {
        p.stat_partial = h->partial;
        p.stat_limit = pl.interval_rows;
      }
      rc = launch_back(h, p, nb, sb);
      if (rc) break;
      if (h->fused_valid) {
        if (p.stat_partial) {
          const uint64_t room = pl.interval_rows > h->pow_rows ? pl.interval_rows - h->pow_rows : 0;
          h->fused_rows += std::min<uint64_t>(rows, room);
        } else {
          h->fused_valid = false;   // this launch ran a kernel that does not accumulate
        }
      }
      h->pow_rows += rows;
      while (!rc && !fused_ok(h) && h->pow_rows >= pl.interval_rows) rc = finalize_interval(h, pl.interval_rows, d_out, cap, rows_written, sb, ch);
      if (rc) break;
    }
    chain_back_done(ch, sb);
    if (ch->ln || outer) {
      dev_event_record(h->region_ev[reg], sb);
      h->region_busy[reg] = true;
    }
    h->blocks_done += nb;
  }
  h->stg_cur = nullptr;
  if (ch == &own) chain_end(&own);
  return rc;
}

int engine_flush(frbch_handle* h, uint8_t* d_out, size_t cap, uint64_t* rows_written, dev_stream_t s, Chain* ch = nullptr) {
  *rows_written = 0;
  if (h->deferred.active && !fused_ok(h)) {   // the scan ended inside its first interval: statistics over what there is, then the digitising pass
    int rc = run_stats(h, h->deferred.rows, s);
    if (rc) return rc;
    h->have_scale = true;
    if (h->cfg.rescale_constant) h->scale_frozen = true;
    h->fused_rows = 0;
    h->fused_valid = false;
    return deferred_emit(h, d_out, cap, rows_written, s);
  }
  if (fused_ok(h) || h->pow_rows == 0) return FRBCH_OK;
  if (h->quant_busy) {
    (void)dev_stream_wait(s, h->quant_ev);
    h->quant_busy = false;
  }
  return finalize_interval(h, std::min<uint64_t>(h->pow_rows, h->pl.interval_rows), d_out, cap, rows_written, s, ch);
}

// offset = 0, scale = 1 on the handle's stream.  Not waited for: work that follows on ANOTHER stream is ordered behind an event
// (join_reset); a reset of the 8 handles of a scan used to cost 8 host round trips with the GPU idle.
int set_identity_rescale(frbch_handle* h) {
  const Plan& pl = h->pl;
  CHECK_DEV(h, dev_memset32(h->offset, 0u, pl.ncol, h->stream), "clear offset");
  CHECK_DEV(h, dev_memset32(h->scale, 0x3F800000u, pl.ncol, h->stream), "unit scale");     // 1.0f
  if (!h->reset_ev_made) {
    CHECK_DEV(h, dev_event_create_sync(&h->reset_ev), "hipEventCreate");
    h->reset_ev_made = true;
  }
  dev_event_record(h->reset_ev, h->stream);
  h->reset_pending = true;
  return FRBCH_OK;
}
// the device entry points: stream `s` continues behind the handle's pending reset
void join_reset(frbch_handle* h, dev_stream_t s) {
  if (!h->reset_pending) return;
  if (s != h->stream) (void)dev_stream_wait(s, h->reset_ev);
  h->reset_pending = false;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
// Work the device entry points queued on a caller's stream (statistics writing offset / scale, the digitiser reading
// them, the power buffer) must be complete before the handle's own stream or the host touches that state.
// The caller's stream may be gone by then (a temporary stream of the caller's framework): what is waited for is an event of
// the handle's own, recorded on that stream behind the call's work (mark_user_stream).
static int settle_user_stream(frbch_handle* h) {
  if (h->user_stream) {
    h->user_stream = 0;                     // (cleared first: a failing wait must not wedge every later call)
    if (h->user_ev_made && dev_event_sync(h->user_ev) != 0)
      return fail(h, FRBCH_E_DEVICE, std::string("sync (caller stream): ") + dev_last_error_string());
  }
  return FRBCH_OK;
}
static void mark_user_stream(frbch_handle* h, dev_stream_t s) {
  if (!h->user_ev_made) {
    if (dev_event_create_sync(&h->user_ev) != 0) return;
    h->user_ev_made = true;
  }
  dev_event_record(h->user_ev, s);
  h->user_stream = s;
}

#ifdef FRBCH_EXPERIMENTS
extern "C" const char* frbch_version(void) { return "frbch abi 4 backend " FRBCH_BACKEND_NAME " +experiments"; }
#else
extern "C" const char* frbch_version(void) { return "frbch abi 4 backend " FRBCH_BACKEND_NAME; }
#endif

extern "C" int frbch_open(const frbch_config* cfg, frbch_handle** out) {
  if (!cfg || !out) return FRBCH_E_ARG;
  *out = nullptr;
  if (cfg->size != sizeof(frbch_config) || cfg->abi_version != FRBCH_ABI_VERSION) return FRBCH_E_ARG;
  frbch_handle* h = new frbch_handle();
  h->cfg = *cfg;
  *out = h;  // returned even on failure so that frbch_last_error works; caller closes it
  if (cfg->device < 0) return fail(h, FRBCH_E_DEVICE, "device < 0: there is no CPU fallback");
  const int ndev = dev_count();
  if (ndev <= 0) return fail(h, FRBCH_E_DEVICE, "no GPU visible to HIP (there is no CPU fallback)");
  if (cfg->device >= ndev) return fail(h, FRBCH_E_DEVICE, "device ordinal out of range");
  if (cfg->flags & ~kAcceptedFlags)
    return fail(h, FRBCH_E_ARG, "unknown bit in cfg.flags (kernel variants kept for A/B runs and timing-only ablations exist only in FRBCH_EXPERIMENTS builds)");
  if (!kExperiments) {   // overlap: automatic (0), off (1), or the plain-stream mode 3 with a CU count; no CU-masked lanes, no forced batching
    const uint32_t mode = (cfg->overlap >> 24) & 0xFFu, batches = (cfg->overlap >> 16) & 0xFFu;
    if ((mode != 0 && mode != 3) || batches)
      return fail(h, FRBCH_E_ARG, "cfg.overlap: CU-masked lane modes and forced batching exist only in FRBCH_EXPERIMENTS builds");
  }
  h->device = cfg->device;
  DeviceGuard dg(h->device);
  char arch[128] = "";
  if (!dev_arch_ok(h->device, arch, sizeof arch, &h->lds_limit)) return fail(h, FRBCH_E_DEVICE, "cannot query device");
  h->lane_ncu = dev_cu_count(h->device);
  const std::string why = make_plan(h->cfg, &h->pl, h->lds_limit);
  if (!why.empty()) return fail(h, FRBCH_E_ARG, why);
  const Plan& pl = h->pl;
  CHECK_DEV(h, dev_stream_create(&h->stream), "hipStreamCreate");
  CHECK_DEV(h, dev_allow_lds(frbch_k1_branch, pl.k1_lds), "LDS size K1");
  CHECK_DEV(h, dev_allow_lds(frbch_k2_chan, pl.k2_lds), "LDS size K2");
  CHECK_DEV(h, dev_allow_lds(frbch_kc_dcfix, pl.kc_lds), "LDS size Kc");

  int rc;
  if ((rc = upload_table(h, &h->tw_r, pl.r, std::max(1, pl.r / 2), 1))) return rc;
  if ((rc = upload_table(h, &h->tw_c2, pl.c2, pl.c, 1))) return rc;
  const uint64_t nlo = 1ull << pl.log2_nlo, nhi = pl.n >> pl.log2_nlo;
  if ((rc = upload_table(h, &h->tw_nlo, pl.n, nlo, 1))) return rc;
  if ((rc = upload_table(h, &h->tw_nhi, pl.n, std::max<uint64_t>(1, nhi), nlo))) return rc;
  h->priv_grid = pl.fast_k2_priv ? 2 * std::max(1, h->lane_ncu) : 0;   // two 80-KiB workgroups per CU
  if ((rc = setup_fast(h))) return rc;
  {
    char nm[64];
    if (pl.fast_k1_log2m && pl.fast_k1_wave) {
      const int nw = pl.fast_k1_kind == 1 ? 4 : (pl.fast_k1_kind == 3 ? 16 : 8), wps = pl.fast_k1_kind == 5 ? 4 : (pl.fast_k1_kind >= 2 ? 2 : 1);   // kind 4: <4,8,2>, kind 5: <5,8,4>
      snprintf(nm, sizeof nm, "frbch_k1_wave<%d,%d,%d>", pl.fast_k1_log2m, nw, wps);
      h->kname[KID_K1] = nm;
    } else if (pl.fast_k1_log2m) {
      snprintf(nm, sizeof nm, "frbch_k1_fast<%d>", pl.fast_k1_log2m);
      h->kname[KID_K1] = nm;
    }
    if (pl.fast_k2_log2m) {
      snprintf(nm, sizeof nm, "frbch_kc_fast<%d>", pl.fast_k2_log2m);
      h->kname[KID_KC] = nm;
      if (pl.fast_k2_wave) {
        const bool two = (pl.fast_k2_log2m == 4) || (pl.fast_k2_log2m == 3 && !(h->cfg.flags & 32u) && pl.fast_k2_nw != 8);
        const int nw = two ? (pl.fast_k2_nw == 2 ? 4 : 8) : pl.fast_k2_nw;
        const int pm = h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? 4 : 0);
        if (pl.fast_k2_log2m == 5) snprintf(nm, sizeof nm, "frbch_k2_wave<5,8,%d,4>", pm);
        else snprintf(nm, sizeof nm, "frbch_k2_wave<%d,%d,%d,%d>", pl.fast_k2_log2m, nw, pm, two ? 2 : 1);
      } else {
        snprintf(nm, sizeof nm, "frbch_k2_fast<%d,%d>", pl.fast_k2_log2m, pl.fast_k2_nt);
      }
      h->kname[KID_K2] = nm;
    } else if (pl.fast_k2_m1) {
      snprintf(nm, sizeof nm, "frbch_k2_wave<0,%d,%d,1>", pl.fast_k2_nw, h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? 4 : 0));
      h->kname[KID_K2] = nm;
    }
#ifndef FRBCH_NO_FAST
    if (pl.fast_k2_priv) {
      snprintf(nm, sizeof nm, "frbch_k2_priv<%d>", h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? 4 : 0));
      h->kname[KID_K2] = nm;
      snprintf(nm, sizeof nm, "frbch_k2_priv<%d,stats>", h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? 4 : 0));
      h->kname[KID_K2S] = nm;
    }
    if (pl.fast_k2_lane) {
      snprintf(nm, sizeof nm, "frbch_k2_lane<%d,%d>", pl.fast_k2_lane, h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? 4 : 0));
      h->kname[KID_K2] = nm;
      if (pl.fast_k2_lane == 1) h->kname[KID_KC] = "frbch_kc_lane";
    }
#endif
  }

  CHECK_DEV(h, dev_malloc((void**)&h->spill, (size_t)pl.maxb * (pl.c2 / pl.g) * pl.gs * sizeof(cf)), "hipMalloc(spill)");
#ifndef FRBCH_NO_FAST
  if (pl.k2_two_stage)
    CHECK_DEV(h, dev_malloc((void**)&h->scr2, (size_t)pl.maxb * (pl.r / pl.k2_stage1_tscr) * pl.ncol * sizeof(float)), "hipMalloc(tscrunch scratch)");
#endif
  CHECK_DEV(h, dev_malloc((void**)&h->s_dc, (size_t)pl.maxb * pl.c2 * sizeof(cf)), "hipMalloc(s_dc)");
  CHECK_DEV(h, dev_malloc((void**)&h->p0, (size_t)pl.maxb * pl.c2 * sizeof(cf)), "hipMalloc(p0)");
  if (pl.coherent) {
    CHECK_DEV(h, dev_allow_lds(frbch_k2c_chirp, pl.k2_lds), "LDS size K2c");
    CHECK_DEV(h, dev_allow_lds(frbch_k3_dedisp, pl.k3_lds), "LDS size K3");
    CHECK_DEV(h, dev_malloc((void**)&h->spill2, (size_t)pl.maxb * pl.n * sizeof(cf)), "hipMalloc(spill2)");
    CHECK_DEV(h, dev_malloc((void**)&h->chirp, (size_t)pl.n * sizeof(cf)), "hipMalloc(chirp)");
    CHECK_DEV(h, dev_malloc((void**)&h->ptmp, (size_t)pl.maxb * pl.rows_per_block * pl.ncol * sizeof(float)), "hipMalloc(ptmp)");
    if ((rc = build_chirp(h, pl.coh_fast_r ? (1 << pl.coh_fast_r) : 0))) return rc;
    h->kname[KID_K2] = pl.coh_fast_c ? "frbch_k2c_fast" : "frbch_k2c_chirp";
    if (pl.ncol % 64 == 0 && pl.rows_per_block % 2 == 0 && pl.c % 4 == 0 && !(h->cfg.flags & 2u)) h->kname[KID_K4] = "frbch_k4_fast";
    if (pl.coh_fast_r) h->kname[KID_K3] = (pl.coh_fast_r == 4 && pl.coh_nt == 512 && !(h->cfg.flags & 8u)) ? "frbch_k3_wave<4>" : "frbch_k3_fast";
  }
  CHECK_DEV(h, dev_malloc((void**)&h->offset, pl.ncol * sizeof(float)), "hipMalloc(offset)");
  CHECK_DEV(h, dev_malloc((void**)&h->scale, pl.ncol * sizeof(float)), "hipMalloc(scale)");
  if ((rc = set_identity_rescale(h))) return rc;
  if (pl.interval_rows == 0) {  // -I0: no rescale, digitise the raw power
    h->have_scale = true;
    h->scale_frozen = true;
  }
  return FRBCH_OK;
}

extern "C" void frbch_close(frbch_handle* h) {
  if (!h) return;
  DeviceGuard dg(h->device);
  (void)settle_user_stream(h);
  if (h->stream) (void)dev_sync(h->stream);
  drain_events(h);
  dev_free(h->tw_r); dev_free(h->tw_c2); dev_free(h->tw_nhi); dev_free(h->tw_nlo);
  dev_free(h->ftw1_h); dev_free(h->ftw2_h);
  dev_free(h->ftw1_r); dev_free(h->ftw2_r); dev_free(h->ftw1_c); dev_free(h->ftw2_c); dev_free(h->td1); dev_free(h->td2);
  dev_free(h->spill); dev_free(h->s_dc); dev_free(h->p0);
  dev_free(h->spill2); dev_free(h->chirp); dev_free(h->ptmp); dev_free(h->scr2);
  dev_free(h->offset); dev_free(h->scale); dev_free(h->powbuf); dev_free(h->partial);
  dev_free(h->d_frames); dev_free(h->d_out); dev_free(h->stg); dev_free(h->d_fbad); dev_free(h->scan_rows);
  for (int i = 0; i < 8; ++i) { dev_host_free(h->pin_in[i]); dev_host_free(h->pin_out[i]); }
  if (h->stream) dev_stream_destroy(h->stream);
  if (h->user_ev_made) dev_event_destroy(h->user_ev);
  if (h->quant_ev_made) dev_event_destroy(h->quant_ev);
  if (h->reset_ev_made) dev_event_destroy(h->reset_ev);
  if (h->region_ev_made) for (auto& e : h->region_ev) dev_event_destroy(e);
  for (auto& e : h->evpool) dev_event_destroy(e);
  delete h;
}

extern "C" const char* frbch_last_error(frbch_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int frbch_get_info(frbch_handle* h, frbch_info* info) {
  if (!h || !info) return FRBCH_E_ARG;
  const Plan& pl = h->pl;
  memset(info, 0, sizeof *info);
  info->size = (uint32_t)sizeof *info;
  info->nchan = pl.c; info->freq_res = pl.r; info->tscrunch = pl.tscr; info->nif = pl.nif;
  info->block_samples = pl.n;
  info->block_payload_bytes = pl.block_payload_bytes;
  info->rows_per_block = pl.rows_per_block;
  info->row_bytes = pl.row_bytes;
  info->rescale_interval_rows = pl.interval_rows;
  info->rows_out = h->rows_out;
  info->blocks_done = h->blocks_done;
  info->tsamp_s = pl.tsamp_s;
  info->tstart_mjd = h->tstart_mjd;
  info->fch1_mhz = pl.fch1; info->foff_mhz = pl.foff;
  info->frame_bytes = h->have_vdif ? h->v0.frame_bytes : 0;
  info->header_bytes = h->have_vdif ? h->v0.header_bytes() : 0;
  info->have_rescale = h->have_scale ? 1 : 0;
  info->diag = h->diag;
  info->frames_seen = h->frames_seen;
  info->frames_invalid = h->frames_invalid;
  info->frame_gaps = h->frame_gaps;
  info->frames_filled = h->frames_filled;
  info->block_stride_bytes = pl.block_stride_bytes;
  info->nfilt_pos = (uint32_t)pl.nfilt_pos;
  info->nfilt_neg = (uint32_t)pl.nfilt_neg;
  return FRBCH_OK;
}

extern "C" int frbch_reset(frbch_handle* h) {
  if (!h) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  { const int rc = settle_user_stream(h); if (rc) return rc; }
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  h->pow_rows = 0;
  h->fused_rows = 0;
  h->fused_valid = false;
  h->deferred.active = false;
  h->rows_out = h->blocks_done = 0;
  h->have_vdif = false;
  h->frames_seen = h->frames_invalid = h->frame_gaps = h->frames_filled = 0;
  h->carry_bad.clear();
  h->next_frame_index = 0;
  h->checked_bytes = 0;
  h->carry.clear();
  h->outq.clear();
  h->outq_pos = 0;
  h->skip_bytes = 0;
  // (the staging areas stay: hipMalloc of the gigabytes a four-product interval needs costs ~0.25 s per handle, which a
  // scan of 8 IFs paid in every call -- 1.9 of 2.2 s, profiles/r03_scan_host_path.txt; stream_begin re-uses them)
  const bool off = h->pl.interval_rows == 0;
  h->have_scale = off;
  h->scale_frozen = off;
  return set_identity_rescale(h);
}

extern "C" int frbch_get_rescale(frbch_handle* h, float* offset, float* scale) {
  if (!h || !offset || !scale) return FRBCH_E_ARG;
  if (!h->have_scale) return fail(h, FRBCH_E_STATE, "rescale not measured yet");
  DeviceGuard dg(h->device);
  { const int rc = settle_user_stream(h); if (rc) return rc; }
  CHECK_DEV(h, dev_d2h(offset, h->offset, h->pl.ncol * sizeof(float), h->stream), "download offset");
  CHECK_DEV(h, dev_d2h(scale, h->scale, h->pl.ncol * sizeof(float), h->stream), "download scale");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  return FRBCH_OK;
}

extern "C" int frbch_set_rescale(frbch_handle* h, const float* offset, const float* scale) {
  if (!h || !offset || !scale) return FRBCH_E_ARG;
  if (h->pow_rows || h->deferred.active) return fail(h, FRBCH_E_STATE, "set_rescale while an interval is being measured");
  DeviceGuard dg(h->device);
  { const int rc = settle_user_stream(h); if (rc) return rc; }
  CHECK_DEV(h, dev_h2d(h->offset, offset, h->pl.ncol * sizeof(float), h->stream), "upload offset");
  CHECK_DEV(h, dev_h2d(h->scale, scale, h->pl.ncol * sizeof(float), h->stream), "upload scale");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  h->have_scale = true;
  h->scale_frozen = true;
  return FRBCH_OK;
}

extern "C" int frbch_process_device(frbch_handle* h, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                                    uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nblocks,
                                    void* d_out, size_t out_cap_bytes, uint64_t* rows_written, void* stream) {
  if (!h || !d_frames || !rows_written || frame_bytes <= header_bytes) return FRBCH_E_ARG;
  const uint64_t payload = (uint64_t)nframes * (frame_bytes - header_bytes);
  if (nblocks && payload_byte_offset + (nblocks - 1) * h->pl.block_stride_bytes + h->pl.block_payload_bytes > payload)
    return fail(h, FRBCH_E_ARG, "frames do not cover the requested blocks");
  if (nblocks && !d_out) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  dev_stream_t s = stream ? (dev_stream_t)stream : h->stream;
  if (stream) {
    if (h->user_stream && h->user_stream != s) { const int rc = settle_user_stream(h); if (rc) return rc; }
  }
  join_reset(h, s);
  const int rc = engine_feed(h, (const uint8_t*)d_frames, frame_bytes, header_bytes, payload_byte_offset, nblocks,
                             (uint8_t*)d_out, out_cap_bytes, rows_written, s);
  if (stream) mark_user_stream(h, s);
  return rc;
}

extern "C" int frbch_flush_device(frbch_handle* h, void* d_out, size_t out_cap_bytes, uint64_t* rows_written,
                                  void* stream) {
  if (!h || !rows_written) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  dev_stream_t s = stream ? (dev_stream_t)stream : h->stream;
  if (stream) {
    if (h->user_stream && h->user_stream != s) { const int rc = settle_user_stream(h); if (rc) return rc; }
  }
  join_reset(h, s);
  const int rc = engine_flush(h, (uint8_t*)d_out, out_cap_bytes, rows_written, s);
  if (stream) mark_user_stream(h, s);
  return rc;
}

extern "C" int frbch_scan_device(frbch_handle* const* ifs, uint32_t nif, const void* const* d_frames, size_t nframes,
                                 uint32_t frame_bytes, uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nblocks,
                                 int flush, void* d_rows, size_t row_pitch_bytes, uint64_t rows_cap, uint64_t* rows_written,
                                 void* stream) {
  if (!ifs || !nif || !ifs[0] || !rows_written || frame_bytes <= header_bytes) return FRBCH_E_ARG;
  frbch_handle* h0 = ifs[0];
  *rows_written = 0;
  const Plan& pl = h0->pl;
  for (uint32_t i = 0; i < nif; ++i) {
    if (!ifs[i] || (nblocks && (!d_frames || !d_frames[i]))) return fail(h0, FRBCH_E_ARG, "null handle or frame pointer in the scan");
    const Plan& a = ifs[i]->pl;
    if (a.c != pl.c || a.nif != pl.nif || a.tscr != pl.tscr || a.row_bytes != pl.row_bytes || ifs[i]->cfg.nbit_out != h0->cfg.nbit_out ||
        ifs[i]->device != h0->device || a.block_stride_bytes != pl.block_stride_bytes || a.block_payload_bytes != pl.block_payload_bytes)
      return fail(h0, FRBCH_E_ARG, "the IFs of a scan must share device, nchan, freq_res, tscrunch, nbit and products");
    // one chain serves all IFs: its mode, lane size and eligibility are taken from ifs[0], so the others must agree
    if (ifs[i]->cfg.overlap != h0->cfg.overlap || ifs[i]->cfg.flags != h0->cfg.flags || a.in_bits != pl.in_bits || a.coherent != pl.coherent)
      return fail(h0, FRBCH_E_ARG, "the IFs of a scan must share flags, overlap, input bits and the coherent setting");
  }
  if (row_pitch_bytes != (size_t)nif * pl.row_bytes) return fail(h0, FRBCH_E_ARG, "row_pitch_bytes must be nif * row_bytes of one IF");
  if (!d_rows) return fail(h0, FRBCH_E_ARG, "null row buffer");
  const uint64_t payload = (uint64_t)nframes * (frame_bytes - header_bytes);
  if (nblocks && payload_byte_offset + (nblocks - 1) * pl.block_stride_bytes + pl.block_payload_bytes > payload)
    return fail(h0, FRBCH_E_ARG, "frames do not cover the requested blocks");
  DeviceGuard dg(h0->device);
  dev_stream_t s = stream ? (dev_stream_t)stream : h0->stream;
  for (uint32_t i = 0; i < nif; ++i) {
    frbch_handle* h = ifs[i];
    if (h->user_stream && h->user_stream != s) { const int rc = settle_user_stream(h); if (rc) return rc; }
    if (!stream && h != h0) CHECK_DEV(h0, dev_sync(h->stream), "sync");   // (earlier work of this IF on its own stream)
    join_reset(h, s);
  }
  // one chain over all IFs: the front stages of IF i + 1 overlap the back stages (and the flush) of IF i
  Lanes* ln = overlap_usable(h0) ? get_lanes(h0->device, overlap_front_cus(h0), overlap_mode(h0) == 3) : nullptr;
  uint64_t stages = 0;
  for (uint32_t i = 0; i < nif; ++i) stages += feed_stage_count(ifs[i], nblocks, ln != nullptr && overlap_mode(h0) == 1);
  Chain ch;
  chain_begin(&ch, h0, s, ln, (uint32_t)stages, pl.nif < 4, overlap_mode(h0));
  const size_t seg = pl.row_bytes / pl.nif;                     // bytes of one product line of one IF
  const uint64_t bits = pl.row_bytes * 8 / pl.ncol;
  uint64_t rows_min = UINT64_MAX;
  int rc = FRBCH_OK;
  for (uint32_t i = 0; i < nif && !rc; ++i) {
    frbch_handle* h = ifs[i];
    h->out_pitch = (uint64_t)nif * pl.c;                        // values per (row, product) line of the scan's rows
    uint8_t* dst = (uint8_t*)d_rows + (size_t)i * seg;
    const size_t cap = (size_t)(rows_cap * row_pitch_bytes) - (size_t)i * seg;
    uint64_t r1 = 0, r2 = 0;
    if (nblocks) rc = engine_feed(h, (const uint8_t*)d_frames[i], frame_bytes, header_bytes, payload_byte_offset, nblocks, dst, cap, &r1, s,
                                  nullptr, 0, nullptr, &ch);
    if (!rc && flush) {
      const dev_stream_t sb = (ch.ln && ch.backs) ? ch.s_back : s;
      const uint64_t used = r1 * out_row_span(h);
      rc = engine_flush(h, dst + used, cap - (size_t)used, &r2, sb, &ch);
      if (!rc && r2) chain_back_touch(&ch);
    }
    h->out_pitch = 0;
    if (rc && h != h0) fail(h0, rc, std::string("IF ") + std::to_string(i) + ": " + h->err);
    rows_min = std::min(rows_min, r1 + r2);
    (void)bits;
  }
  chain_end(&ch);
  // every handle whose own stream is not `s` records the scan's work behind it: a later frbch_reset / get_rescale / set_rescale
  // of that handle waits for this event, not only for its own (idle) stream -- with stream == NULL that is every IF but the first
  for (uint32_t i = 0; i < nif; ++i)
    if (stream || ifs[i] != h0) mark_user_stream(ifs[i], s);
  *rows_written = rows_min == UINT64_MAX ? 0 : rows_min;
  return rc;
}

extern "C" int frbch_power_device(frbch_handle* h, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                                  uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nblocks,
                                  float* d_power, size_t cap_bytes, void* stream) {
  if (!h || !d_frames || !d_power || frame_bytes <= header_bytes) return FRBCH_E_ARG;
  const Plan& pl = h->pl;
  const uint64_t payload = (uint64_t)nframes * (frame_bytes - header_bytes);
  if (nblocks && payload_byte_offset + (nblocks - 1) * pl.block_stride_bytes + pl.block_payload_bytes > payload)
    return fail(h, FRBCH_E_ARG, "frames do not cover the requested blocks");
  if (nblocks * pl.rows_per_block * pl.ncol * sizeof(float) > cap_bytes)
    return fail(h, FRBCH_E_CAPACITY, "power buffer too small");
  DeviceGuard dg(h->device);
  dev_stream_t s = stream ? (dev_stream_t)stream : h->stream;
  if (stream) {
    if (h->user_stream && h->user_stream != s) { const int rc = settle_user_stream(h); if (rc) return rc; }
  }
  join_reset(h, s);
  for (uint64_t b0 = 0; b0 < nblocks; b0 += pl.maxb) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(pl.maxb, nblocks - b0);
    KParams p = base_params(h);
    p.frames = (const uint8_t*)d_frames;
    p.frame_bytes = frame_bytes;
    p.header_bytes = header_bytes;
    p.payload_bytes = frame_bytes - header_bytes;
    p.payload_off = payload_byte_offset + b0 * pl.block_stride_bytes;
    int rc = launch_front(h, p, nb, s, s);
    if (rc) return rc;
    p.out_mode = FRBCH_OUT_FLOAT_POWER;
    p.power_out = d_power;
    p.row0 = b0 * pl.rows_per_block;
    rc = launch_back(h, p, nb, s);
    if (rc) return rc;
  }
  if (stream) mark_user_stream(h, s);
  return FRBCH_OK;
}

// the unpack tap: voltages as the filterbank sees them (A4 in isolation)
extern "C" int frbch_unpack_device(frbch_handle* h, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                                   uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nsamples, int decoder,
                                   float* d_volt, size_t cap_bytes, void* stream) {
  if (!h || !d_frames || !d_volt || frame_bytes <= header_bytes) return FRBCH_E_ARG;
  const Plan& pl = h->pl;
  const uint64_t spb = 4 / (uint64_t)pl.in_bits;
  const uint64_t payload = (uint64_t)nframes * (frame_bytes - header_bytes);
  if (payload_byte_offset + (nsamples + spb - 1) / spb > payload) return fail(h, FRBCH_E_ARG, "frames do not cover the requested samples");
  if (decoder != 0 && decoder != 1) return fail(h, FRBCH_E_ARG, "decoder must be 0 (generic K1) or 1 (register kernels)");
  if (decoder == 1 && (pl.in_bits != 2 || (nsamples & 1))) return fail(h, FRBCH_E_ARG, "the register kernels decode 2-bit input, two samples per byte");
  const uint64_t need = (decoder ? 4 : 2) * nsamples * sizeof(float);
  if (need > cap_bytes) return fail(h, FRBCH_E_CAPACITY, "voltage buffer too small");
  if (!nsamples) return FRBCH_OK;
  DeviceGuard dg(h->device);
  dev_stream_t s = stream ? (dev_stream_t)stream : h->stream;
  if (stream) {
    if (h->user_stream && h->user_stream != s) { const int rc = settle_user_stream(h); if (rc) return rc; }
  }
  KParams p = base_params(h);
  p.frames = (const uint8_t*)d_frames;
  p.frame_bytes = frame_bytes;
  p.header_bytes = header_bytes;
  p.payload_bytes = frame_bytes - header_bytes;
  p.payload_off = payload_byte_offset;
  p.power_out = d_volt;
  p.row0 = nsamples;
  if (decoder == 0) {
    DEV_LAUNCH(frbch_unpack_tap, (nsamples + 255) / 256, 1, 256, 0, s, p);
  } else {
#ifndef FRBCH_NO_FAST
    hipLaunchKernelGGL(fast::frbch_unpack_tap_fast, dim3((unsigned)((nsamples / 2 + 255) / 256)), dim3(256), 0, s, p);
#else
    return fail(h, FRBCH_E_ARG, "the register kernels are not part of this build");
#endif
  }
  CHECK_DEV(h, dev_check_launch(), "launch unpack tap");
  if (stream) mark_user_stream(h, s);
  return FRBCH_OK;
}

// ---- profiling ---------------------------------------------------------------------------------
extern "C" int frbch_set_profiling(frbch_handle* h, int enable) {
  if (!h) return FRBCH_E_ARG;
  h->profiling = enable != 0;
  return FRBCH_OK;
}
extern "C" int frbch_timing_reset(frbch_handle* h) {
  if (!h) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  (void)settle_user_stream(h);
  (void)dev_sync(h->stream);
  drain_events(h);
  for (int i = 0; i < KID_COUNT; ++i) h->acc_ms[i] = h->acc_bytes[i] = 0.0, h->acc_launches[i] = 0;
  return FRBCH_OK;
}
extern "C" int frbch_get_timing(frbch_handle* h, frbch_timing* t) {
  if (!h || !t) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  drain_events(h);
  memset(t, 0, sizeof *t);
  t->size = (uint32_t)sizeof *t;
  t->nkernels = KID_COUNT;
  for (int i = 0; i < KID_COUNT; ++i) {
    snprintf(t->k[i].name, sizeof t->k[i].name, "%s", h->kname[i].empty() ? kKernelNames[i] : h->kname[i].c_str());
    t->k[i].launches = h->acc_launches[i];
    t->k[i].total_ms = h->acc_ms[i];
    t->k[i].algorithmic_bytes = h->acc_bytes[i];
  }
  return FRBCH_OK;
}

// ---- host streaming -----------------------------------------------------------------------------
namespace {

int stream_begin(frbch_handle* h, const uint8_t* first_frame) {
  VdifInfo v;
  if (!parse_vdif_header(first_frame, &v)) return fail(h, FRBCH_E_FORMAT, "not a VDIF frame header");
  std::string why;
  if (!check_vdif_supported(v, &why)) return fail(h, FRBCH_E_FORMAT, "unsupported VDIF stream: " + why);
  h->v0 = v;
  h->have_vdif = true;
  if ((int)v.bits_per_sample != h->pl.in_bits) {
    if (h->cfg.input_bits) return fail(h, FRBCH_E_FORMAT, "VDIF bits/sample differs from cfg.input_bits");
    const std::string why2 = make_plan(h->cfg, &h->pl, h->lds_limit, (int)v.bits_per_sample);   // same sizes, other gather
    if (!why2.empty()) return fail(h, FRBCH_E_ARG, why2);
    CHECK_DEV(h, dev_allow_lds(frbch_k1_branch, h->pl.k1_lds), "LDS size K1");
    CHECK_DEV(h, dev_allow_lds(frbch_k2_chan, h->pl.k2_lds), "LDS size K2");
    h->kname[KID_K1].clear();
    dev_free(h->spill);   // the group size (hence the padded slab count) may have changed
    h->spill = nullptr;
    CHECK_DEV(h, dev_malloc((void**)&h->spill, (size_t)h->pl.maxb * (h->pl.c2 / h->pl.g) * h->pl.gs * sizeof(cf)), "hipMalloc(spill)");
  }
  const Plan& pl = h->pl;
  const uint64_t spb = 4 / (uint64_t)pl.in_bits;                       // dual-pol samples per payload byte
  const double fps = pl.rate_in * 2.0 * pl.in_bits / 8.0 / v.payload_bytes();
  uint64_t s0 = (uint64_t)llround(h->cfg.start_s * pl.rate_in);
  s0 -= s0 % spb;
  h->skip_bytes = s0 / spb;
  const double want = h->cfg.total_s * pl.rate_in;
  // blocks start every `hop` samples and read N: whole blocks inside the first -T seconds
  const uint64_t nwant = want >= 9.0e18 ? UINT64_MAX : (uint64_t)llround(want);
  h->blocks_budget = nwant == UINT64_MAX ? UINT64_MAX : (nwant >= pl.n ? (nwant - pl.n) / pl.hop + 1 : 0);
  // overlap-save drops the first nfilt_pos channel samples (2C input samples each) of the stream
  h->tstart_mjd = (double)vdif_epoch_mjd((int)v.ref_epoch) +
                  ((double)v.seconds + (double)v.frame_nr / fps +
                   (double)(s0 + (uint64_t)pl.c2 * (uint64_t)pl.nfilt_pos) / pl.rate_in) / 86400.0;
  // device staging: frames of one launch batch, output of one batch (+ a completed interval)
  const uint64_t pb = v.payload_bytes();
  const uint64_t nfr = ((uint64_t)(pl.maxb - 1) * pl.block_stride_bytes + pl.block_payload_bytes + pb - 1) / pb + 2;
  const size_t want_frames = (size_t)(nfr * v.frame_bytes);
  if (!h->d_frames || h->d_frames_cap < want_frames) {
    dev_free(h->d_frames);
    h->d_frames = nullptr;
    h->d_frames_cap = want_frames;
    CHECK_DEV(h, dev_malloc((void**)&h->d_frames, h->d_frames_cap), "hipMalloc(frame staging)");
  }
  // scan mode: the rows go straight into the scan's row buffer (out_target); the staging area only carries the backlog of
  // an IF that is ahead of the others (a few batches), never a whole interval
  const uint64_t burst_rows = (h->sink ? 0 : pl.interval_rows) + 2ull * pl.maxb * pl.rows_per_block;
  const size_t want_out = (size_t)(burst_rows * pl.row_bytes);
  if (!h->d_out || h->d_out_cap < want_out) {
    dev_free(h->d_out);
    h->d_out = nullptr;
    h->d_out_cap = want_out;
    CHECK_DEV(h, dev_malloc((void**)&h->d_out, h->d_out_cap), "hipMalloc(output staging)");
  }
  return FRBCH_OK;
}

// Where the rows of the next engine call of the host paths go: the handle's own staging area (packed rows), or -- scan mode --
// straight into this IF's columns of the scan's row buffer, behind the rows it already holds: the frequency concatenation
// happens in the store addresses of K2 / the digitiser (KParams::out_pitch), no copy.
struct OutTarget { uint8_t* ptr; size_t cap; };
OutTarget out_target(frbch_handle* h) {
  if (!h->sink) {
    h->out_pitch = 0;
    return OutTarget{h->d_out, h->d_out_cap};
  }
  const Plan& pl = h->pl;
  const uint64_t bits = pl.row_bytes * 8 / pl.ncol;
  h->out_pitch = (uint64_t)h->sink_line_pitch * 8 / bits;        // values per (row, product) line of the scan's rows
  const size_t row_pitch = h->sink_line_pitch * pl.nif;
  const uint64_t room = h->sink_rows_cap > h->sink_rows ? h->sink_rows_cap - h->sink_rows : 0;
  return OutTarget{h->sink + h->sink_rows * row_pitch, room ? (size_t)((room - 1) * row_pitch + (size_t)(pl.nif - 1) * h->sink_line_pitch + pl.row_bytes / pl.nif) : 0};
}

int queue_rows(frbch_handle* h, uint64_t rows) {
  if (!rows) return FRBCH_OK;
  if (h->sink) {   // scan mode: the rows are already in this IF's columns of the scan buffer (out_target)
    if (h->sink_rows + rows > h->sink_rows_cap) return fail(h, FRBCH_E_CAPACITY, "scan row buffer too small");
    CHECK_DEV(h, dev_sync(h->stream), "sync");
    h->sink_rows += rows;
    return FRBCH_OK;
  }
  const size_t nbytes = rows * h->pl.row_bytes;
  if (h->outq_pos && h->outq_pos == h->outq.size()) {
    h->outq.clear();
    h->outq_pos = 0;
  }
  const size_t old = h->outq.size();
  h->outq.resize(old + nbytes);
  CHECK_DEV(h, dev_d2h(h->outq.data() + old, h->d_out, nbytes, h->stream), "download rows");
  CHECK_DEV(h, dev_sync(h->stream), "sync");
  return FRBCH_OK;
}

// every frame header of the stream is checked once: geometry must match the first frame (else the
// stream is corrupt or mis-framed: error); frames flagged invalid are counted and marked (their samples enter the
// filterbank as 0), a forward jump of the frame number is filled with zero-valued filler frames so that the stream stays
// contiguous in time (-cont, process_vdif.py:157,160), a backward jump is only counted
int check_headers(frbch_handle* h) {
  const uint64_t fb = h->v0.frame_bytes;
  const double fps_d = h->pl.rate_in * 2.0 * h->pl.in_bits / 8.0 / h->v0.payload_bytes();
  const uint64_t fps = (uint64_t)llround(fps_d);
  // a frame is looked at once its whole header is there (a filler copies header_bytes() of it: 32 for non-legacy VDIF)
  while (h->checked_bytes + h->v0.header_bytes() <= h->carry.size()) {
    VdifInfo v;
    parse_vdif_header(h->carry.data() + h->checked_bytes, &v);
    if (v.frame_bytes != h->v0.frame_bytes || v.legacy != h->v0.legacy || v.bits_per_sample != h->v0.bits_per_sample ||
        v.log2_nchan != h->v0.log2_nchan)
      return fail(h, FRBCH_E_FORMAT, "VDIF frame header changes geometry mid-stream (frame " + std::to_string(h->frames_seen) + ")");
    const uint64_t idx = (uint64_t)v.seconds * fps + v.frame_nr;
    if (h->frames_seen && idx != h->next_frame_index) {
      h->frame_gaps++;
      if (idx > h->next_frame_index) {   // frames are missing: keep the stream contiguous in time with zero-valued fillers
        const uint64_t nfill = idx - h->next_frame_index;
        if (nfill > 16 * fps) return fail(h, FRBCH_E_FORMAT, "VDIF frame numbers jump forward by more than 16 s (frame " + std::to_string(h->frames_seen) + ")");
        std::vector<uint8_t> filler((size_t)(nfill * fb), 0);
        for (uint64_t k = 0; k < nfill; ++k) {
          memcpy(filler.data() + k * fb, h->carry.data() + h->checked_bytes, h->v0.header_bytes());
          filler[k * fb + 3] |= 0x80;      // invalid flag (word 0, bit 31)
        }
        h->carry.insert(h->carry.begin() + (long)h->checked_bytes, filler.begin(), filler.end());
        h->carry_bad.resize(h->checked_bytes / fb, 0);
        h->carry_bad.insert(h->carry_bad.end(), (size_t)nfill, 1);
        h->frames_filled += nfill;
        h->checked_bytes += nfill * fb;
      }
      // (a backward jump -- duplicate or re-ordered frames -- is counted and the data used as they come: -cont)
    }
    h->next_frame_index = idx + 1;
    if (v.invalid) h->frames_invalid++;
    h->carry_bad.resize(h->checked_bytes / fb, 0);
    h->carry_bad.push_back(v.invalid ? 1 : 0);
    h->frames_seen++;
    h->checked_bytes += fb;
  }
  return FRBCH_OK;
}

// bitmap of the bad-frame flags of `nfr` frames starting at flags[0], on the device (null when none is set)
int upload_bad_frames(frbch_handle* h, const uint8_t* flags, uint64_t nfr, const uint32_t** d_out) {
  *d_out = nullptr;
  bool any = false;
  for (uint64_t f = 0; f < nfr && !any; ++f) any = flags[f] != 0;
  if (!any) return FRBCH_OK;
  const size_t words = (size_t)((nfr + 31) / 32);
  if (h->d_fbad_words < words) {
    dev_free(h->d_fbad);
    h->d_fbad = nullptr;
    CHECK_DEV(h, dev_malloc((void**)&h->d_fbad, words * sizeof(uint32_t)), "hipMalloc(frame flags)");
    h->d_fbad_words = words;
  }
  std::vector<uint32_t> bits(words, 0u);
  for (uint64_t f = 0; f < nfr; ++f)
    if (flags[f]) bits[f >> 5] |= 1u << (f & 31);
  CHECK_DEV(h, dev_h2d(h->d_fbad, bits.data(), words * sizeof(uint32_t), h->stream), "upload frame flags");
  CHECK_DEV(h, dev_sync(h->stream), "sync");     // `bits` is released on return
  *d_out = h->d_fbad;
  return FRBCH_OK;
}

int process_carry(frbch_handle* h) {
  const Plan& pl = h->pl;
  const uint64_t fb = h->v0.frame_bytes, hb = h->v0.header_bytes(), pb = h->v0.payload_bytes();
  {
    const int rc = check_headers(h);
    if (rc) return rc;
  }
  size_t consumed_frames = 0;  // frames at the front of carry that are fully used
  for (;;) {
    // drop frames that -S skips entirely
    const uint64_t frames_avail = (h->carry.size() - consumed_frames * fb) / fb;
    const uint64_t drop = std::min<uint64_t>(h->skip_bytes / pb, frames_avail);
    consumed_frames += drop;
    h->skip_bytes -= drop * pb;
    const uint64_t fa = frames_avail - drop;
    if (fa * pb < h->skip_bytes + pl.block_payload_bytes || h->blocks_budget == 0) break;
    uint64_t nb = (fa * pb - h->skip_bytes - pl.block_payload_bytes) / pl.block_stride_bytes + 1;
    nb = std::min<uint64_t>(nb, std::min<uint64_t>(pl.maxb, h->blocks_budget));
    const uint64_t need_frames = (h->skip_bytes + (nb - 1) * pl.block_stride_bytes + pl.block_payload_bytes + pb - 1) / pb;
    const uint8_t* src = h->carry.data() + consumed_frames * fb;
    CHECK_DEV(h, dev_h2d(h->d_frames, src, need_frames * fb, h->stream), "upload frames");
    uint64_t rows = 0;
    const uint8_t* bad = h->carry_bad.size() >= consumed_frames + need_frames ? h->carry_bad.data() + consumed_frames : nullptr;
    const uint32_t* d_bad = nullptr;
    int rc = bad ? upload_bad_frames(h, bad, need_frames, &d_bad) : FRBCH_OK;
    if (rc) return rc;
    const OutTarget ot = out_target(h);
    rc = engine_feed(h, h->d_frames, (uint32_t)fb, (uint32_t)hb, h->skip_bytes, nb, ot.ptr, ot.cap,
                     &rows, h->stream, d_bad ? bad : nullptr, need_frames, d_bad);
    if (rc) return rc;
    rc = queue_rows(h, rows);  // also synchronises, so `src` may be released
    if (rc) return rc;
    if (!rows) CHECK_DEV(h, dev_sync(h->stream), "sync");
    h->blocks_budget -= nb;
    h->skip_bytes += nb * pl.block_stride_bytes;
  }
  if (consumed_frames) {
    h->carry_bad.erase(h->carry_bad.begin(), h->carry_bad.begin() + (long)std::min<size_t>(consumed_frames, h->carry_bad.size()));
    h->carry.erase(h->carry.begin(), h->carry.begin() + consumed_frames * fb);
    h->checked_bytes = h->checked_bytes > consumed_frames * fb ? h->checked_bytes - consumed_frames * fb : 0;
  }
  if (h->blocks_budget == 0) {  // -T reached: ignore the rest
    h->carry_bad.clear();
    h->carry.clear();
    h->checked_bytes = 0;
  }
  return FRBCH_OK;
}

}  // namespace

extern "C" int frbch_push(frbch_handle* h, const uint8_t* frames, size_t nbytes) {
  if (!h || (!frames && nbytes)) return FRBCH_E_ARG;
  if (h->have_vdif && h->blocks_budget == 0) return FRBCH_OK;
  DeviceGuard dg(h->device);
  h->carry.insert(h->carry.end(), frames, frames + nbytes);
  if (!h->have_vdif) {
    if (h->carry.size() < 32) return FRBCH_OK;
    const int rc = stream_begin(h, h->carry.data());
    if (rc) return rc;
  }
  return process_carry(h);
}

extern "C" int frbch_flush(frbch_handle* h) {
  if (!h) return FRBCH_E_ARG;
  if (!h->have_vdif) return FRBCH_OK;
  DeviceGuard dg(h->device);
  uint64_t rows = 0;
  const OutTarget ot = out_target(h);
  int rc = engine_flush(h, ot.ptr, ot.cap, &rows, h->stream);
  if (rc) return rc;
  return queue_rows(h, rows);
}

extern "C" long frbch_pull(frbch_handle* h, uint8_t* dst, size_t cap) {
  if (!h || (!dst && cap)) return FRBCH_E_ARG;
  const size_t avail = h->outq.size() - h->outq_pos;
  const size_t n = std::min(avail, cap);
  if (n) memcpy(dst, h->outq.data() + h->outq_pos, n);
  h->outq_pos += n;
  return (long)n;
}

extern "C" long frbch_sigproc_header(frbch_handle* h, uint8_t* dst, size_t cap) {
  if (!h) return FRBCH_E_ARG;
  if (!h->have_vdif) return fail(h, FRBCH_E_STATE, "no VDIF frame seen yet");
  const std::vector<uint8_t> hdr = sigproc_header(h->cfg, h->pl, h->tstart_mjd);
  if (hdr.size() > cap) return fail(h, FRBCH_E_CAPACITY, "header buffer too small");
  memcpy(dst, hdr.data(), hdr.size());
  return (long)hdr.size();
}

namespace {
bool write_all(int fd, const uint8_t* p, size_t n) {
  while (n) {
    const ssize_t w = write(fd, p, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
  }
  return true;
}
}  // namespace

namespace {

// Whole-file path for a regular input file: a reader thread preads the frames of the next batch into one of two pinned
// buffers while the GPU works on the current one, rows come back through two pinned buffers that a writer thread
// drains into the output (strictly sequential writes: the target may be a FIFO).  No copy through `carry` / `outq`.
// Returns FRBCH_OK, an error, or 1 = "not applicable, use the generic stream path" (nothing consumed).
constexpr int kMaxSlots = 8;
struct PipeQueue {          // ring of pinned slots between the engine thread and the I/O threads
  std::mutex m;
  std::condition_variable cv;
  int ready[kMaxSlots] = {0};    // slot state: 0 free, 1 filled
  size_t nbytes[kMaxSlots] = {0};
  uint64_t offs[kMaxSlots] = {0};   // output: file offset of the slot's bytes (regular files: positional writes)
  bool stop = false;
  int error = 0;
  uint64_t seq_off = 0;          // output: every byte below this file offset has been written (writers that cannot use the
  std::vector<std::pair<uint64_t, size_t>> done_off;   // mapping take turns in file order); pieces finished out of order
};
bool pwrite_all(int fd, const uint8_t* p, size_t n, uint64_t off) {
  while (n) {
    const ssize_t w = pwrite(fd, p, n, (off_t)off);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += w;
    n -= (size_t)w;
    off += (uint64_t)w;
  }
  return true;
}

// hs: one handle (rows come out of its own staging area) or the IFs of a scan (d_rows != null: every handle's rows go
// into its columns of the pitched buffer d_rows through its sink, rows every IF has delivered are written).
// The whole-file fast path addresses frames by their position in the file: it needs a file whose frame numbers run without
// a jump.  First and last header tell (seconds * fps + frame number must advance by exactly the frame count); a file
// with missing frames takes the stream path, which fills the gaps (check_headers).
bool vdif_file_contiguous(int fd, const frbch_config& cfg) {
  struct stat st;
  uint8_t a[16], b[16];
  if (fstat(fd, &st) != 0 || st.st_size < 32 || pread(fd, a, 16, 0) != 16) return true;   // (let the caller report it)
  VdifInfo v0, v1;
  if (!parse_vdif_header(a, &v0) || v0.frame_bytes == 0) return true;
  const uint64_t nfile = (uint64_t)st.st_size / v0.frame_bytes;
  if (nfile < 2 || pread(fd, b, 16, (off_t)((nfile - 1) * v0.frame_bytes)) != 16) return true;
  parse_vdif_header(b, &v1);
  const double fps_d = 2.0e6 * fabs(cfg.bw_mhz) * 2.0 * v0.bits_per_sample / 8.0 / (double)v0.payload_bytes();
  const uint64_t fps = (uint64_t)llround(fps_d);
  const uint64_t i0 = (uint64_t)v0.seconds * fps + v0.frame_nr, i1 = (uint64_t)v1.seconds * fps + v1.frame_nr;
  return i1 - i0 == nfile - 1;
}

#ifdef FRBCH_EXPERIMENTS
struct PhaseClock {   // FRBCH_TIMING=1: wall-clock phases of a whole-file call on stderr
  bool on = getenv("FRBCH_TIMING") != nullptr;
  double t0 = now();
  static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
  void mark(const char* what) { if (on) { const double t = now(); fprintf(stderr, "[frbch timing] %-28s %8.2f ms\n", what, (t - t0) * 1e3); t0 = t; } }
};
#define PHASE_MARK(pc, what) (pc).mark(what)
#else
struct PhaseClock {};
#define PHASE_MARK(pc, what) ((void)0)
#endif

int run_pipelined(frbch_handle* const* hs, uint32_t nif, const int* in_fds, int out_fd, uint8_t* d_rows, size_t row_pitch) {
  frbch_handle* h0 = hs[0];
  PhaseClock pc;
  (void)pc;
  const bool scan = d_rows != nullptr;
  struct Batch { uint64_t nb, f0, nfr, pay_off; };
  std::vector<std::vector<Batch>> batches(nif);
  uint64_t fbmax = 0;
  for (uint32_t i = 0; i < nif; ++i) {
    frbch_handle* h = hs[i];
    struct stat st;
    if (fstat(in_fds[i], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 32 || h->have_vdif) return i == 0 ? 1 : fail(h0, FRBCH_E_IO, "scan inputs must all be regular files");
    uint8_t first[32];
    if (pread(in_fds[i], first, 32, 0) != 32) return i == 0 ? 1 : fail(h0, FRBCH_E_IO, "short read");
    const int rc0 = stream_begin(h, first);
    if (rc0) return h == h0 ? rc0 : fail(h0, rc0, std::string("IF ") + std::to_string(i) + ": " + h->err);
    const Plan& pl = h->pl;
    const uint64_t fb = h->v0.frame_bytes, pb = h->v0.payload_bytes();
    fbmax = std::max(fbmax, fb);
    const uint64_t nfile = (uint64_t)st.st_size / fb;                       // whole frames in the file
    const uint64_t pay_total = nfile * pb;
    uint64_t nblk = 0;
    if (pay_total >= h->skip_bytes + pl.block_payload_bytes)
      nblk = (pay_total - h->skip_bytes - pl.block_payload_bytes) / pl.block_stride_bytes + 1;
    nblk = std::min<uint64_t>(nblk, h->blocks_budget);
    const uint64_t nbatch = (nblk + pl.maxb - 1) / pl.maxb;
    const uint64_t per = nbatch ? (nblk + nbatch - 1) / nbatch : 1;       // equal batches
    for (uint64_t b0 = 0; b0 < nblk; b0 += per) {
      Batch b;
      b.nb = std::min<uint64_t>(per, nblk - b0);
      const uint64_t p0 = h->skip_bytes + b0 * pl.block_stride_bytes;
      const uint64_t p1 = p0 + (b.nb - 1) * pl.block_stride_bytes + pl.block_payload_bytes;
      b.f0 = p0 / pb;
      b.nfr = (p1 + pb - 1) / pb - b.f0;
      b.pay_off = p0 - b.f0 * pb;
      batches[i].push_back(b);
    }
  }
  const Plan& pl0 = h0->pl;
  int rc = FRBCH_OK;

  PHASE_MARK(pc, "stream_begin (allocations)");
  // Pinned rings (kept in the first handle: pinning costs ~0.5 ms per MB).  Page-cache / tmpfs reads run at ~3 GB/s per
  // thread, far below the PCIe link, and scale with threads: each input slot has its own reader thread (pread).
  // Writes into ONE file do not scale (the kernel serialises them per inode; positional writes from four threads and
  // parallel copies into a shared mapping were both measured slower than one stream, tools_runfile_timing.py): one
  // writer, strictly sequential write() calls -- which is also what a FIFO target needs (base2fil.sh:348-349).
  const size_t slot_bytes = 16u << 20;
  const size_t in_cap = (size_t)std::max<uint64_t>(fbmax, (slot_bytes / fbmax) * fbmax);
  const size_t out_cap = std::max<size_t>(slot_bytes, scan ? row_pitch : 0);
  uint64_t in_total = 0;
  for (uint32_t i = 0; i < nif; ++i)
    for (const Batch& b : batches[i]) in_total += b.nfr * hs[i]->v0.frame_bytes;
  const int NR = (int)std::min<uint64_t>(kMaxSlots, std::max<uint64_t>(2, (in_total + in_cap - 1) / in_cap));
  const int NSO = NR;                                   // output slots
  // A regular output file of known size is preallocated (posix_fallocate, in a thread of its own while the input is
  // being read) and then written through a shared mapping by one writer per slot: the pages exist, the writers' copies
  // only map them (minor faults run in parallel; measured on tmpfs: fallocate 29 ms + 8 copying threads 27 ms for 319 MB,
  // against 75 ms for write() calls and 110+ ms for copies that have to allocate the pages in their faults).  Anything
  // else -- FIFOs (base2fil.sh:348-349), character devices -- keeps ONE writer and strictly sequential write() calls.
  struct stat ost;
  const bool out_regular = fstat(out_fd, &ost) == 0 && S_ISREG(ost.st_mode);
  uint64_t out_expect = 0;
  if (out_regular) {
    uint64_t rows_min = UINT64_MAX;
    for (uint32_t i = 0; i < nif; ++i) {
      uint64_t nb = 0;
      for (const Batch& b : batches[i]) nb += b.nb;
      rows_min = std::min<uint64_t>(rows_min, nb * hs[i]->pl.rows_per_block);
    }
    const size_t hdr_bytes = sigproc_header(h0->cfg, h0->pl, 0.0, scan ? (int)(h0->pl.c * nif) : 0).size();
    out_expect = hdr_bytes + rows_min * (scan ? (uint64_t)row_pitch : h0->pl.row_bytes);
    if (out_expect < (8u << 20)) out_expect = 0;        // small outputs: not worth a mapping
  }
  uint8_t* omap = nullptr;                              // set by the preallocation thread
  std::atomic<int> pre_done{out_expect ? 0 : 1};
  std::thread prealloc;
  const int NWR = out_expect ? NSO : 1;                 // writer threads (each owns one slot when the output is mapped)
  if (h0->pin_in_cap < in_cap || h0->pin_out_cap < out_cap) {
    for (int i = 0; i < kMaxSlots; ++i) {
      dev_host_free(h0->pin_in[i]); h0->pin_in[i] = nullptr;
      dev_host_free(h0->pin_out[i]); h0->pin_out[i] = nullptr;
    }
    h0->pin_in_cap = in_cap;
    h0->pin_out_cap = out_cap;
  }
  uint8_t** inbuf = h0->pin_in;
  uint8_t** outbuf = h0->pin_out;
  for (int i = 0; i < NR; ++i)
    if ((!inbuf[i] && dev_host_alloc((void**)&inbuf[i], in_cap) != 0) || (!outbuf[i] && dev_host_alloc((void**)&outbuf[i], out_cap) != 0))
      return fail(h0, FRBCH_E_NOMEM, "pinned staging buffers");
  auto release = [&]() {};   // (the rings stay with the handle until frbch_close)
  // (started only now: no early return is left between here and the join at the end of this function)
  h0->diag &= ~1u;
  if (out_expect)
    prealloc = std::thread([&]() {
#ifdef FRBCH_TEST_HOOKS
      const bool refuse = getenv("FRBCH_TEST_NO_MMAP") != nullptr;   // (emulator build only: exercises the writers' fallback)
#else
      const bool refuse = false;
#endif
      if (!refuse && posix_fallocate(out_fd, 0, (off_t)out_expect) == 0) {
        void* m = mmap(nullptr, (size_t)out_expect, PROT_READ | PROT_WRITE, MAP_SHARED, out_fd, 0);   // needs a descriptor opened O_RDWR (open_output)
        if (m != MAP_FAILED) omap = (uint8_t*)m;
      }
      pre_done.store(1, std::memory_order_release);    // (without a mapping the writers take turns: one sequential stream of write() calls)
    });

  PHASE_MARK(pc, "pinned rings");
  // every batch's frames travel in pieces of whole frames that fit a pinned buffer; batch rounds go IF by IF
  struct Piece { uint32_t ifx; size_t batch; uint64_t f0, nfr; size_t dst_off; bool ends_batch, ends_round; };
  std::vector<Piece> pieces;
  size_t nround = 0;
  for (uint32_t i = 0; i < nif; ++i) nround = std::max(nround, batches[i].size());
  for (size_t r = 0; r < nround; ++r) {
    for (uint32_t i = 0; i < nif; ++i) {
      if (r >= batches[i].size()) continue;
      const Batch& b = batches[i][r];
      const uint64_t fb = hs[i]->v0.frame_bytes;
      const uint64_t per = in_cap / fb;
      for (uint64_t f = 0; f < b.nfr; f += per)
        pieces.push_back(Piece{i, r, b.f0 + f, std::min<uint64_t>(per, b.nfr - f), (size_t)(f * fb), f + per >= b.nfr, false});
    }
    if (!pieces.empty()) pieces.back().ends_round = true;
  }

  PipeQueue qin, qout;
  std::vector<std::thread> readers, writers;
  for (int r = 0; r < NR; ++r)
    readers.emplace_back([&, r]() {                // reader r owns input slot r: pieces r, r + NR, ...
      for (size_t i = (size_t)r; i < pieces.size(); i += (size_t)NR) {
        {
          std::unique_lock<std::mutex> lk(qin.m);
          qin.cv.wait(lk, [&] { return qin.ready[r] == 0 || qin.stop; });
          if (qin.stop) return;
        }
        const uint64_t fb = hs[pieces[i].ifx]->v0.frame_bytes;
        const size_t want = (size_t)(pieces[i].nfr * fb);
        size_t got = 0;
        int err = 0;
        while (got < want) {
          const ssize_t n = pread(in_fds[pieces[i].ifx], inbuf[r] + got, want - got, (off_t)(pieces[i].f0 * fb + got));
          if (n < 0 && errno == EINTR) continue;
          if (n <= 0) { err = n < 0 ? errno : EIO; break; }
          got += (size_t)n;
        }
        std::lock_guard<std::mutex> lk(qin.m);
        if (got != want) qin.error = err ? err : EIO;
        qin.nbytes[r] = got;
        qin.ready[r] = 1;
        qin.cv.notify_all();
        if (qin.error) return;
      }
    });
  for (int w = 0; w < NWR; ++w)
    writers.emplace_back([&, w]() {                // mapped output: writer w owns slot w; else one writer, slots in order
      for (size_t i = (size_t)w;; i += (size_t)NWR) {
        const int slot = (int)(i % (size_t)NSO);
        size_t n;
        uint64_t off;
        {
          std::unique_lock<std::mutex> lk(qout.m);
          qout.cv.wait(lk, [&] { return qout.ready[slot] == 1 || qout.stop; });
          if (qout.ready[slot] != 1) return;       // stop and nothing pending
          n = qout.nbytes[slot];
          off = qout.offs[slot];
        }
        bool ok = true;
        if (NWR > 1) {
          while (!pre_done.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
          if (omap && off + n <= out_expect) {
            memcpy(omap + off, outbuf[slot], n);
          } else {
            // no mapping (or more bytes than expected): the writers take turns in file order -- ONE sequential stream of
            // writes, never concurrent positional writes into one inode (measured slower than a single writer)
            {
              std::unique_lock<std::mutex> lk(qout.m);
              qout.cv.wait(lk, [&] { return qout.seq_off == off || qout.error; });
            }
            ok = pwrite_all(out_fd, outbuf[slot], n, off);
          }
          {
            std::lock_guard<std::mutex> lk(qout.m);
            if (qout.seq_off == off) qout.seq_off = off + n;    // (mapped copies advance it too, so that a later unmapped slot finds its turn)
            else qout.done_off.push_back({off, n});
            for (bool moved = true; moved;) {
              moved = false;
              for (size_t k = 0; k < qout.done_off.size(); ++k)
                if (qout.done_off[k].first == qout.seq_off) {
                  qout.seq_off += qout.done_off[k].second;
                  qout.done_off.erase(qout.done_off.begin() + (long)k);
                  moved = true;
                  break;
                }
            }
            qout.cv.notify_all();
          }
        } else {
          ok = write_all(out_fd, outbuf[slot], n);
        }
        const int err = ok ? 0 : (errno ? errno : EIO);
        std::lock_guard<std::mutex> lk(qout.m);
        if (!ok) qout.error = err;
        qout.ready[slot] = 0;
        qout.cv.notify_all();
        if (!ok) return;
      }
    });
  size_t out_i = 0;                                // next output slot
  uint64_t out_off = 0;                            // bytes handed to the writers so far = file offset of the next slot
  auto out_acquire = [&]() -> int {                // wait until the slot is free; returns slot or -1 on writer error
    const int slot = (int)(out_i % (size_t)NSO);
    std::unique_lock<std::mutex> lk(qout.m);
    qout.cv.wait(lk, [&] { return qout.ready[slot] == 0 || qout.error; });
    return qout.error ? -1 : slot;
  };
  auto out_submit = [&](int slot, size_t n) {
    std::lock_guard<std::mutex> lk(qout.m);
    qout.nbytes[slot] = n;
    qout.offs[slot] = out_off;
    out_off += n;
    qout.ready[slot] = 1;
    ++out_i;
    qout.cv.notify_all();
  };
  // `total` bytes at device address `src` -> pinned slots -> writers.  The copy of slot k+1 is queued before the engine
  // thread waits for slot k, so the link stays busy while a writer drains.
  auto emit_bytes = [&](const uint8_t* src, size_t total, dev_stream_t st) -> int {
    for (size_t off = 0; off < total;) {
      const size_t n = std::min(out_cap, total - off);
      const int slot = out_acquire();
      if (slot < 0) return fail(h0, FRBCH_E_IO, std::string("write: ") + strerror(qout.error));
      CHECK_DEV(h0, dev_d2h(outbuf[slot], src + off, n, st), "download rows");
      CHECK_DEV(h0, dev_sync(st), "sync");
      out_submit(slot, n);
      off += n;
    }
    return FRBCH_OK;
  };
  // scan: write the rows every IF has delivered, keep the rest at the top of the buffer (cf. frbch_run_scan)
  auto drain_scan = [&](bool final_) -> int {
    uint64_t n = UINT64_MAX;
    for (uint32_t i = 0; i < nif; ++i) n = std::min(n, hs[i]->sink_rows);
    if (n && n != UINT64_MAX) {
      for (uint32_t i = 0; i < nif; ++i) CHECK_DEV(h0, dev_sync(hs[i]->stream), "sync");
      const int rc1 = emit_bytes(d_rows, (size_t)(n * row_pitch), h0->stream);
      if (rc1) return rc1;
    }
    if (final_) return FRBCH_OK;
    const size_t seg = pl0.row_bytes / pl0.nif, line_pitch = row_pitch / pl0.nif;
    for (uint32_t i = 0; i < nif; ++i) {
      frbch_handle* h = hs[i];
      const uint64_t left = h->sink_rows - n;
      if (left && n) {
        if (left * h->pl.row_bytes > h->d_out_cap) return fail(h0, FRBCH_E_CAPACITY, "scan backlog exceeds the staging area");
        CHECK_DEV(h0, dev_copy2d(h->d_out, seg, h->sink + n * pl0.nif * line_pitch, line_pitch, seg, left * pl0.nif, h->stream), "move backlog");
        CHECK_DEV(h0, dev_copy2d(h->sink, line_pitch, h->d_out, seg, seg, left * pl0.nif, h->stream), "move backlog");
        CHECK_DEV(h0, dev_sync(h->stream), "sync");
      }
      h->sink_rows = left;
    }
    return FRBCH_OK;
  };

  {   // SIGPROC header first
    if (scan)
      for (uint32_t i = 1; i < nif && !rc; ++i)
        if (fabs(hs[i]->tstart_mjd - h0->tstart_mjd) > 0.5 * pl0.tsamp_s / 86400.0)
          rc = fail(h0, FRBCH_E_FORMAT, "the IFs of the scan do not start at the same time");
    const std::vector<uint8_t> hdr = sigproc_header(h0->cfg, pl0, h0->tstart_mjd, scan ? (int)(pl0.c * nif) : 0);
    const int slot = rc ? -1 : out_acquire();
    if (slot < 0) { if (!rc) rc = fail(h0, FRBCH_E_IO, "write"); }
    else {
      memcpy(outbuf[slot], hdr.data(), hdr.size());
      out_submit(slot, hdr.size());
    }
  }
  std::vector<uint64_t> checked_upto(nif, 0);      // file frame index below which headers were checked
  std::vector<std::vector<uint8_t>> bad_all(nif);  // per IF and file frame: 1 = flagged invalid (its samples read as 0)
  for (uint32_t i = 0; i < nif; ++i) {
    uint64_t last = 0;
    for (const Batch& b : batches[i]) last = std::max(last, b.f0 + b.nfr);
    bad_all[i].assign((size_t)last, 0);
  }
  for (size_t pi = 0; pi < pieces.size() && !rc; ++pi) {
    const int slot = (int)(pi % (size_t)NR);
    const Piece& pc = pieces[pi];
    frbch_handle* h = hs[pc.ifx];
    const Plan& pl = h->pl;
    const uint64_t fb = h->v0.frame_bytes, hb = h->v0.header_bytes(), pb = h->v0.payload_bytes();
    {
      std::unique_lock<std::mutex> lk(qin.m);
      qin.cv.wait(lk, [&] { return qin.ready[slot] == 1; });
      if (qin.error) {
        rc = fail(h0, FRBCH_E_IO, std::string("read: ") + strerror(qin.error));
        break;
      }
    }
    // header checks (frames shared with the previous batch are not counted twice): as check_headers
    const uint64_t fps = (uint64_t)llround(pl.rate_in * 2.0 * pl.in_bits / 8.0 / (double)pb);
    for (uint64_t f = std::max(pc.f0, checked_upto[pc.ifx]); f < pc.f0 + pc.nfr && !rc; ++f) {
      VdifInfo v;
      parse_vdif_header(inbuf[slot] + (f - pc.f0) * fb, &v);
      if (v.frame_bytes != h->v0.frame_bytes || v.legacy != h->v0.legacy || v.bits_per_sample != h->v0.bits_per_sample ||
          v.log2_nchan != h->v0.log2_nchan)
        rc = fail(h0, FRBCH_E_FORMAT, "VDIF frame header changes geometry mid-stream (frame " + std::to_string(f) + ")");
      const uint64_t idx = (uint64_t)v.seconds * fps + v.frame_nr;
      if (h->frames_seen && idx != h->next_frame_index) h->frame_gaps++;
      h->next_frame_index = idx + 1;
      if (v.invalid) {
        h->frames_invalid++;
        if (f < bad_all[pc.ifx].size()) bad_all[pc.ifx][(size_t)f] = 1;
      }
      h->frames_seen++;
    }
    checked_upto[pc.ifx] = std::max(checked_upto[pc.ifx], pc.f0 + pc.nfr);
    if (rc) break;
    if (dev_h2d(h->d_frames + pc.dst_off, inbuf[slot], (size_t)(pc.nfr * fb), h->stream) != 0 || dev_sync(h->stream) != 0) {
      rc = fail(h0, FRBCH_E_DEVICE, std::string("upload frames: ") + dev_last_error_string());
      break;
    }
    {   // the pinned buffer is free again: the reader may fill it with the piece after next
      std::lock_guard<std::mutex> lk(qin.m);
      qin.ready[slot] = 0;
      qin.cv.notify_all();
    }
    if (pc.ends_batch) {
      const Batch& b = batches[pc.ifx][pc.batch];
      uint64_t rows = 0;
      const uint8_t* bad = bad_all[pc.ifx].data() + b.f0;
      const uint32_t* d_bad = nullptr;
      rc = upload_bad_frames(h, bad, b.nfr, &d_bad);
      const OutTarget ot = out_target(h);
      if (!rc) rc = engine_feed(h, h->d_frames, (uint32_t)fb, (uint32_t)hb, b.pay_off, b.nb, ot.ptr, ot.cap, &rows, h->stream,
                                d_bad ? bad : nullptr, b.nfr, d_bad);
      if (rc && h != h0) fail(h0, rc, std::string("IF ") + std::to_string(pc.ifx) + ": " + h->err);
      if (!rc && rows) rc = scan ? queue_rows(h, rows) : emit_bytes(h->d_out, (size_t)(rows * pl.row_bytes), h->stream);
      h->blocks_budget -= std::min<uint64_t>(h->blocks_budget, b.nb);
      h->skip_bytes += b.nb * pl.block_stride_bytes;
    }
    if (!rc && scan && pc.ends_round) rc = drain_scan(false);
  }
  PHASE_MARK(pc, "input + transform");
  for (uint32_t i = 0; i < nif && !rc; ++i) {
    frbch_handle* h = hs[i];
    uint64_t rows = 0;
    const OutTarget ot = out_target(h);
    rc = engine_flush(h, ot.ptr, ot.cap, &rows, h->stream);
    if (rc && h != h0) fail(h0, rc, std::string("IF ") + std::to_string(i) + ": " + h->err);
    if (!rc && rows) rc = scan ? queue_rows(h, rows) : emit_bytes(h->d_out, (size_t)(rows * h->pl.row_bytes), h->stream);
  }
  if (!rc && scan) rc = drain_scan(true);
  PHASE_MARK(pc, "flush + output");
  {   // stop the threads: readers may be waiting for a slot, writers for data
    { std::lock_guard<std::mutex> lk(qin.m); qin.stop = true; qin.cv.notify_all(); }
    for (auto& t : readers) t.join();
    {   // let the writers drain what is queued, then stop
      std::unique_lock<std::mutex> lk(qout.m);
      qout.cv.wait(lk, [&] {
        bool idle = true;
        for (int i = 0; i < NSO; ++i) idle = idle && qout.ready[i] == 0;
        return idle || qout.error;
      });
      qout.stop = true;
      qout.cv.notify_all();
    }
    for (auto& t : writers) t.join();
    if (!rc && qout.error) rc = fail(h0, FRBCH_E_IO, std::string("write: ") + strerror(qout.error));
  }
  if (prealloc.joinable()) prealloc.join();
  if (omap) {
    munmap(omap, (size_t)out_expect);
    h0->diag |= 1u;            // frbch_info::diag bit 0: the output went through the shared mapping
  }
  if (out_expect && out_off != out_expect && ftruncate(out_fd, (off_t)out_off) != 0 && !rc)
    rc = fail(h0, FRBCH_E_IO, std::string("ftruncate: ") + strerror(errno));
  PHASE_MARK(pc, "writers drained");
  release();
  return rc;
}

// The output of a whole-file call: INSTALL.md:32-35 -- no O_EXCL, so that a pre-made FIFO (base2fil.sh:348-349) can be the
// target; never unlinked, never seeked.  A regular file (or a new one) is opened O_RDWR: the preallocated shared mapping of
// run_pipelined needs read access to the descriptor (mmap of a write-only descriptor fails with EACCES); FIFOs, devices and
// files that only grant write access keep O_WRONLY and ONE sequential writer.
int open_output(const char* path) {
  struct stat st;
  const bool special = stat(path, &st) == 0 && !S_ISREG(st.st_mode);
  if (!special) {
    const int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd >= 0 || (errno != EACCES && errno != EPERM)) return fd;
  }
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
#ifdef F_SETPIPE_SZ
  if (fd >= 0 && special && S_ISFIFO(st.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);   // a FIFO: 1 MiB instead of 64 KiB in flight (best effort)
#endif
  return fd;
}

int run_file_pipelined(frbch_handle* h, int in_fd, int out_fd) {
  frbch_handle* hs[1] = {h};
  return run_pipelined(hs, 1, &in_fd, out_fd, nullptr, 0);
}

}  // namespace

extern "C" int frbch_run_file(frbch_handle* h, const char* vdif_path, const char* out_fil) {
  if (!h || !vdif_path || !out_fil) return FRBCH_E_ARG;
  DeviceGuard dg(h->device);
  if (!(h->cfg.flags & kFlagNoPipeline)) {   // regular input file: overlapped read / transform / write
    const int in_fd = open(vdif_path, O_RDONLY);
    if (in_fd < 0) return fail(h, FRBCH_E_IO, std::string("cannot open ") + vdif_path + ": " + strerror(errno));
    struct stat st;
    if (fstat(in_fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 32 && !h->have_vdif && vdif_file_contiguous(in_fd, h->cfg)) {
      const int out_fd = open_output(out_fil);
      if (out_fd < 0) {
        close(in_fd);
        return fail(h, FRBCH_E_IO, std::string("cannot open ") + out_fil + ": " + strerror(errno));
      }
      int rc = run_file_pipelined(h, in_fd, out_fd);
      close(in_fd);
      if (close(out_fd) != 0 && !rc) rc = fail(h, FRBCH_E_IO, std::string("close: ") + strerror(errno));
      if (rc != 1) return rc;
    } else {
      close(in_fd);
    }
  }
  FILE* in = fopen(vdif_path, "rb");
  if (!in) return fail(h, FRBCH_E_IO, std::string("cannot open ") + vdif_path + ": " + strerror(errno));
  // INSTALL.md:32-35: no O_EXCL, so that a pre-made FIFO (base2fil.sh:348-349) can be the target
  const int fd = open(out_fil, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd < 0) {
    fclose(in);
    return fail(h, FRBCH_E_IO, std::string("cannot open ") + out_fil + ": " + strerror(errno));
  }
  int rc = FRBCH_OK;
  std::vector<uint8_t> buf(32u << 20), out(8u << 20);
  bool header_done = false;
  auto drain = [&]() -> int {
    if (!header_done && h->have_vdif) {
      const long n = frbch_sigproc_header(h, out.data(), out.size());
      if (n < 0) return (int)n;
      if (!write_all(fd, out.data(), (size_t)n)) return fail(h, FRBCH_E_IO, std::string("write: ") + strerror(errno));
      header_done = true;
    }
    for (;;) {
      const long n = frbch_pull(h, out.data(), out.size());
      if (n < 0) return (int)n;
      if (n == 0) return FRBCH_OK;
      if (!write_all(fd, out.data(), (size_t)n)) return fail(h, FRBCH_E_IO, std::string("write: ") + strerror(errno));
    }
  };
  // -S: skip whole frames by seeking (the engine skips the remainder inside the first frame)
  for (;;) {
    const size_t n = fread(buf.data(), 1, buf.size(), in);
    if (n == 0) break;
    if ((rc = frbch_push(h, buf.data(), n))) break;
    if ((rc = drain())) break;
    if (h->have_vdif && h->blocks_budget == 0) break;
  }
  if (!rc) rc = frbch_flush(h);
  if (!rc) rc = drain();
  if (!rc && !header_done) rc = fail(h, FRBCH_E_FORMAT, "input holds no complete VDIF frame");
  fclose(in);
  if (close(fd) != 0 && !rc) rc = fail(h, FRBCH_E_IO, std::string("close: ") + strerror(errno));
  return rc;
}

// =============================================================================================
// One scan, several IFs on one GPU: replaces N digifil processes + N FIFOs + splice
// =============================================================================================
static constexpr size_t kScanPushBytes = 32u << 20;   // bytes of one IF pushed between two drains of the fallback loop
extern "C" int frbch_run_scan(frbch_handle* const* ifs, uint32_t nif, const char* const* vdif_paths, const char* out_fil) {
  if (!ifs || !nif || !vdif_paths || !out_fil || !ifs[0]) return FRBCH_E_ARG;
  frbch_handle* h0 = ifs[0];
  for (uint32_t i = 0; i < nif; ++i) {
    if (!ifs[i] || !vdif_paths[i]) return fail(h0, FRBCH_E_ARG, "null handle or path in the scan");
    const Plan &a = ifs[i]->pl, &b = h0->pl;
    if (a.c != b.c || a.nif != b.nif || a.tscr != b.tscr || a.row_bytes != b.row_bytes || ifs[i]->cfg.nbit_out != h0->cfg.nbit_out ||
        ifs[i]->device != h0->device || a.tsamp_s != b.tsamp_s)
      return fail(h0, FRBCH_E_ARG, "the IFs of a scan must share device, nchan, tscrunch, nbit and products");
    if (ifs[i]->have_vdif) return fail(h0, FRBCH_E_STATE, "frbch_run_scan needs freshly opened (or reset) handles");
  }
  DeviceGuard dg(h0->device);
  const Plan& pl = h0->pl;
  const size_t seg = pl.row_bytes / pl.nif, line_pitch = seg * nif, row_pitch = line_pitch * pl.nif;
  // one completed interval + the batches one round can deliver: the pipelined path drains after every batch round (3
  // batches of slack); the push-driven fallback pushes kScanPushBytes per IF between drains, which small blocks turn
  // into several batches
  uint64_t push_batches = 0;
  for (uint32_t i = 0; i < nif; ++i) {
    const Plan& q = ifs[i]->pl;
    push_batches = std::max<uint64_t>(push_batches, kScanPushBytes / std::max<uint64_t>(1, (uint64_t)q.maxb * q.block_stride_bytes) + 2);
  }
  const uint64_t rows_cap = pl.interval_rows + std::max<uint64_t>(3, push_batches) * pl.maxb * pl.rows_per_block + 16;
  // the scan's row buffer stays with the first handle (tens of GB for a four-product interval of 8 IFs: allocating and
  // freeing it cost ~1 s per call, profiles/r03_scan_host_path.txt)
  if (!h0->scan_rows || h0->scan_rows_bytes < rows_cap * row_pitch) {
    dev_free(h0->scan_rows);
    h0->scan_rows = nullptr;
    h0->scan_rows_bytes = rows_cap * row_pitch;
    CHECK_DEV(h0, dev_malloc((void**)&h0->scan_rows, h0->scan_rows_bytes), "hipMalloc(scan rows)");
  }
  uint8_t* const d_rows = h0->scan_rows;
  std::vector<FILE*> in(nif, nullptr);
  int fd = -1, rc = FRBCH_OK;
  uint8_t* stage = nullptr;            // pinned host staging of finished rows
  const size_t stage_bytes = 64u << 20;
  auto cleanup = [&]() {
    for (FILE* f : in) if (f) fclose(f);
    if (fd >= 0) close(fd);
    dev_host_free(stage);
    for (uint32_t i = 0; i < nif; ++i) { ifs[i]->sink = nullptr; ifs[i]->out_pitch = 0; }
  };
  for (uint32_t i = 0; i < nif && !rc; ++i) {
    in[i] = fopen(vdif_paths[i], "rb");
    if (!in[i]) rc = fail(h0, FRBCH_E_IO, std::string("cannot open ") + vdif_paths[i] + ": " + strerror(errno));
    ifs[i]->sink = d_rows + (size_t)i * seg;
    ifs[i]->sink_line_pitch = line_pitch;
    ifs[i]->sink_rows = 0;
    ifs[i]->sink_rows_cap = rows_cap;
  }
  if (!rc) {
    fd = open_output(out_fil);   // no O_EXCL: may be a FIFO (INSTALL.md:32-35)
    if (fd < 0) rc = fail(h0, FRBCH_E_IO, std::string("cannot open ") + out_fil + ": " + strerror(errno));
  }
  if (!rc && !(h0->cfg.flags & kFlagNoPipeline)) {   // regular input files: overlapped read / transform / write (run_pipelined)
    std::vector<int> fds(nif, -1);
    bool regular = true;
    for (uint32_t i = 0; i < nif; ++i) {
      fds[i] = open(vdif_paths[i], O_RDONLY);
      struct stat st;
      if (fds[i] < 0 || fstat(fds[i], &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 32 || !vdif_file_contiguous(fds[i], ifs[i]->cfg)) regular = false;
    }
    if (regular) {
      rc = run_pipelined(ifs, nif, fds.data(), fd, d_rows, row_pitch);
      for (int f : fds) if (f >= 0) close(f);
      const int fd1 = fd;
      fd = -1;
      if (fd1 >= 0 && close(fd1) != 0 && !rc) rc = fail(h0, FRBCH_E_IO, std::string("close: ") + strerror(errno));
      cleanup();
      return rc;
    }
    for (int f : fds) if (f >= 0) close(f);
  }
  if (!rc && dev_host_alloc((void**)&stage, stage_bytes) != 0) rc = fail(h0, FRBCH_E_NOMEM, "pinned staging buffer");
  bool header_done = false;
  // write the rows every IF has delivered; rows only some IFs have stay in the buffer (moved to its top)
  auto drain = [&](bool final_) -> int {
    uint64_t n = UINT64_MAX, most = 0;
    for (uint32_t i = 0; i < nif; ++i) {
      if (!ifs[i]->have_vdif) return FRBCH_OK;      // nothing can be written before every IF has started
      n = std::min(n, ifs[i]->sink_rows);
      most = std::max(most, ifs[i]->sink_rows);
    }
    if (!header_done) {
      for (uint32_t i = 1; i < nif; ++i)
        if (fabs(ifs[i]->tstart_mjd - h0->tstart_mjd) > 0.5 * pl.tsamp_s / 86400.0)
          return fail(h0, FRBCH_E_FORMAT, "the IFs of the scan do not start at the same time");
      const std::vector<uint8_t> hdr = sigproc_header(h0->cfg, pl, h0->tstart_mjd, (int)(pl.c * nif));
      if (!write_all(fd, hdr.data(), hdr.size())) return fail(h0, FRBCH_E_IO, std::string("write: ") + strerror(errno));
      header_done = true;
    }
    for (uint64_t r0 = 0; r0 < n;) {
      const uint64_t nr = std::min<uint64_t>(n - r0, stage_bytes / row_pitch);
      if (!nr) return fail(h0, FRBCH_E_CAPACITY, "row larger than the staging buffer");
      CHECK_DEV(h0, dev_d2h(stage, d_rows + r0 * row_pitch, nr * row_pitch, h0->stream), "download scan rows");
      CHECK_DEV(h0, dev_sync(h0->stream), "sync");
      if (!write_all(fd, stage, nr * row_pitch)) return fail(h0, FRBCH_E_IO, std::string("write: ") + strerror(errno));
      r0 += nr;
    }
    if (final_) return FRBCH_OK;                      // cut to the shortest IF, as splice does
    for (uint32_t i = 0; i < nif; ++i) {
      frbch_handle* h = ifs[i];
      const uint64_t left = h->sink_rows - n;
      if (left && n) {                                // via this IF's own staging area: the two regions may overlap
        if (left * h->pl.row_bytes > h->d_out_cap) return fail(h0, FRBCH_E_CAPACITY, "scan backlog exceeds the staging area");
        CHECK_DEV(h0, dev_copy2d(h->d_out, seg, h->sink + n * pl.nif * line_pitch, line_pitch, seg, left * pl.nif, h->stream), "move backlog");
        CHECK_DEV(h0, dev_copy2d(h->sink, line_pitch, h->d_out, seg, seg, left * pl.nif, h->stream), "move backlog");
        CHECK_DEV(h0, dev_sync(h->stream), "sync");
      }
      h->sink_rows = left;
    }
    (void)most;
    return FRBCH_OK;
  };
  std::vector<uint8_t> buf(kScanPushBytes);
  std::vector<bool> eof(nif, false);
  while (!rc) {
    bool any = false;
    for (uint32_t i = 0; i < nif && !rc; ++i) {
      frbch_handle* h = ifs[i];
      if (eof[i] || (h->have_vdif && h->blocks_budget == 0)) continue;
      const size_t n = fread(buf.data(), 1, buf.size(), in[i]);
      if (!n) { eof[i] = true; continue; }
      any = true;
      if ((rc = frbch_push(h, buf.data(), n)) && h != h0) fail(h0, rc, std::string("IF ") + std::to_string(i) + ": " + h->err);
    }
    if (!rc) rc = drain(false);
    if (!any) break;
  }
  for (uint32_t i = 0; i < nif && !rc; ++i)
    if ((rc = frbch_flush(ifs[i])) && ifs[i] != h0) fail(h0, rc, std::string("IF ") + std::to_string(i) + ": " + ifs[i]->err);
  if (!rc) rc = drain(true);
  if (!rc && !header_done) rc = fail(h0, FRBCH_E_FORMAT, "an input holds no complete VDIF frame");
  const int fd_ = fd;
  fd = -1;
  if (fd_ >= 0 && close(fd_) != 0 && !rc) rc = fail(h0, FRBCH_E_IO, std::string("close: ") + strerror(errno));
  cleanup();
  return rc;
}

#include "frbch_post.inc"
