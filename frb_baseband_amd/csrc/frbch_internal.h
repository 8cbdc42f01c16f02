// libfrbch compute side, shared by its translation units (round 4: the former single 3 000-line frbch_engine.cpp):
//   frbch_launch.cpp    plan -> kernel launches: every HIP kernel is instantiated and launched here (the only unit that sees the
//                       kernel sources), LDS permissions, tables, statistics / digitiser launches
//   frbch_stream.cpp    stream state of a handle: batches, the rescale-interval state machine (buffered and two-pass forms),
//                       the chain of stages of a scan and the two kernels that may share the chip
//   frbch_api.cpp       the C ABI of include/frbch.h that works on device memory: life cycle, rescale state, device entry points
//   frbch_file.cpp      the C ABI that works on host memory and files: push / pull, the pipelined whole-file and scan paths
//   frbch_post.cpp      behind / in front of the filterbank: dedispersion, fold, corner turn
// Compiled as HIP for gfx950 (FRBCH_DEV_HEADER = "dev_hip.h").  The CPU unit tests compile the same files against
// tests/emu/dev_emu.h to exercise the host logic without a GPU; that build is test infrastructure and is never loaded by the package.
#ifndef FRBCH_INTERNAL_H
#define FRBCH_INTERNAL_H
#include FRBCH_DEV_HEADER

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
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

using namespace frbch;

enum { KID_K1 = 0, KID_KC, KID_K2, KID_STATS, KID_QUANT, KID_K3, KID_K4, KID_K0, KID_K2S, KID_K2P, KID_COUNT };
const char* const kKernelNames[KID_COUNT] = {"frbch_k1_branch", "frbch_kc_dcfix", "frbch_k2_chan",
                                             "frbch_stats", "frbch_quantise", "frbch_k3_dedisp", "frbch_k4_out", "frbch_k0_stage",
                                             "frbch_k2_statpass", "frbch_k2_priv"};
static_assert(KID_COUNT <= (int)(sizeof(((frbch_timing*)nullptr)->k) / sizeof(((frbch_timing*)nullptr)->k[0])), "frbch_timing holds every slot");

struct EventPair {
  dev_event_t a, b;
  int kid;
  double bytes;
};

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
  float* dls_tab = nullptr;        // dynamic level setting (cfg.unpack_mode 1): [nsample + 1][2] output levels; null = static table
  uint32_t* dls_nlow = nullptr;    // ... low-state counts of the windows of the launch in progress (frbch_dls_count)
  uint64_t dls_cap = 0;            // ... windows it holds
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

namespace frbchi {


inline int fail(frbch_handle* h, int code, const std::string& msg) {
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
constexpr uint32_t kFlagGenericK1 = 1u, kFlagGenericK2 = 2u, kFlagSeparateStats = 1u << 20;
constexpr uint32_t kFlagBuffered = 1u << 27, kFlagTwoPass = 1u << 28;   // form of a first `-c` rescale interval: neither = automatic
constexpr uint32_t kProductFlags = kFlagGenericK1 | kFlagGenericK2 | kFlagSeparateStats | kFlagBuffered | kFlagTwoPass;
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

inline void drain_events(frbch_handle* h) {
  for (auto& e : h->events) {
    h->acc_ms[e.kid] += dev_event_ms(e.a, e.b);
    h->acc_bytes[e.kid] += e.bytes;
    h->acc_launches[e.kid] += 1;
    dev_event_destroy(e.a);
    dev_event_destroy(e.b);
  }
  h->events.clear();
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

// ---- frbch_launch.cpp -----------------------------------------------------------------------------------------------------------
int upload_table(frbch_handle* h, cf** dst, uint64_t n, uint64_t count, uint64_t step);
KParams base_params(const frbch_handle* h);
int allow_generic_lds(frbch_handle* h);                      // LDS sizes of the generic kernels for the handle's plan
int setup_fast(frbch_handle* h);                             // tables and LDS sizes of the register-pass kernels
int build_chirp(frbch_handle* h, int order_m);
int launch_front(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s, dev_stream_t sk);   // K0, K1, Kc
int launch_dls_count(frbch_handle* h, KParams& p, uint64_t nsamples, dev_stream_t s);            // dynamic level setting: the windows' low-state counts
int launch_back(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s);                     // K2 (or K2c, K3, K4)
int launch_unpack_tap(frbch_handle* h, KParams& p, uint64_t nsamples, int decoder, dev_stream_t s);
int fused_chunks_of(const frbch_handle* h);                  // rows of partial rescale sums K2 may write (0: it cannot sum)
int ensure_partial(frbch_handle* h);
int ensure_powbuf(frbch_handle* h);
int run_stats(frbch_handle* h, uint64_t rows, dev_stream_t s);
bool quant_fast_geometry(const frbch_handle* h, int ncu, int wgs_per_cu, uint64_t rp_force, uint64_t* wgs_out, uint64_t* nthr_out, uint64_t* rp_out);
int run_quantise(frbch_handle* h, uint64_t rows, uint8_t* dst, dev_stream_t s, int ncu = 0);
uint64_t out_row_span(const frbch_handle* h);
uint64_t out_extent(const frbch_handle* h, uint64_t rows);
// ---- frbch_stream.cpp -----------------------------------------------------------------------------------------------------------
Lanes* get_lanes(int device, int ncu_front, bool plain = false);
dev_event_t pool_event(frbch_handle* h);
void chain_begin(Chain* c, frbch_handle* owner, dev_stream_t user, Lanes* ln, uint32_t stages_total, bool k0_back, int mode);
void chain_end(Chain* c);
void chain_back_touch(Chain* c);
int overlap_front_cus(const frbch_handle* h);
int overlap_mode(const frbch_handle* h);
bool overlap_usable(const frbch_handle* h);
uint64_t feed_stage_count(const frbch_handle* h, uint64_t nblocks, bool overlap);
int engine_feed(frbch_handle* h, const uint8_t* d_frames, uint32_t frame_bytes, uint32_t header_bytes,
                uint64_t payload_off, uint64_t nblocks, uint8_t* d_out, size_t cap, uint64_t* rows_written,
                dev_stream_t s, const uint8_t* h_bad = nullptr, uint64_t nfr_bad = 0, const uint32_t* d_fbad = nullptr,
                Chain* chain = nullptr);
int engine_flush(frbch_handle* h, uint8_t* d_out, size_t cap, uint64_t* rows_written, dev_stream_t s, Chain* ch = nullptr);
int set_identity_rescale(frbch_handle* h);
void join_reset(frbch_handle* h, dev_stream_t s);
// ---- frbch_api.cpp --------------------------------------------------------------------------------------------------------------
int settle_user_stream(frbch_handle* h);
void mark_user_stream(frbch_handle* h, dev_stream_t s);

}  // namespace frbchi
#endif
