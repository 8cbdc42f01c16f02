// Host-side (no GPU) pieces of libfrbch: .hdr / digifil-argv / VDIF / SIGPROC handling and the
// geometry plan.  Reference anchors are cited per function.
#include "frbch_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <fstream>
#include <sstream>

namespace frbch {

// ---------------------------------------------------------------------------------------------
// VDIF
// ---------------------------------------------------------------------------------------------
static uint32_t rd32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

bool parse_vdif_header(const uint8_t* buf, VdifInfo* v) {
  const uint32_t w0 = rd32(buf), w1 = rd32(buf + 4), w2 = rd32(buf + 8), w3 = rd32(buf + 12);
  v->invalid = (w0 >> 31) & 1;
  v->legacy = (w0 >> 30) & 1;
  v->seconds = w0 & 0x3FFFFFFFu;
  v->ref_epoch = (w1 >> 24) & 0x3F;
  v->frame_nr = w1 & 0xFFFFFFu;
  v->log2_nchan = (w2 >> 24) & 0x1F;
  v->frame_bytes = (w2 & 0xFFFFFFu) * 8u;
  v->is_complex = (w3 >> 31) & 1;
  v->bits_per_sample = ((w3 >> 26) & 0x1F) + 1;
  v->thread_id = (w3 >> 16) & 0x3FF;
  v->station_id = w3 & 0xFFFF;
  return v->frame_bytes > v->header_bytes();
}

bool check_vdif_supported(const VdifInfo& v, std::string* why) {
  std::ostringstream o;
  if (v.bits_per_sample != 2 && v.bits_per_sample != 1) o << "bits/sample = " << v.bits_per_sample << " (need 1 or 2); ";
  if (v.log2_nchan != 1) o << "channels = " << (1u << v.log2_nchan) << " (need 2 = two pols); ";
  if (v.is_complex) o << "complex samples (need real); ";
  if (v.payload_bytes() % 8) o << "payload not a multiple of 8 bytes; ";
  *why = o.str();
  return why->empty();
}

int vdif_epoch_mjd(int ref_epoch) {
  const int year = 2000 + ref_epoch / 2;
  const int month = (ref_epoch % 2 == 0) ? 1 : 7;
  const int a = (14 - month) / 12;
  const int y = year + 4800 - a;
  const int m = month + 12 * a - 3;
  const long jdn = 1 + (153 * m + 2) / 5 + 365L * y + y / 4 - y / 100 + y / 400 - 32045;
  return (int)(jdn - 2400001);
}

// ---------------------------------------------------------------------------------------------
// Plan
// ---------------------------------------------------------------------------------------------
static int ilog2(uint64_t v) {
  int l = 0;
  while ((1ull << l) < v) ++l;
  return l;
}
static bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }
static int pow2_floor(uint64_t v) {
  int p = 1;
  while ((uint64_t)p * 2 <= v) p *= 2;
  return p;
}

std::string coherent_geometry(double freq_mhz, double bw_mhz, uint32_t nchan, uint32_t freq_res, uint32_t tscrunch,
                              double dm, uint32_t* r_out, int* nfilt_pos, int* nfilt_neg) {
  // worst case = the lowest-frequency channel; output sample t needs inputs t - t_hi .. t + t_lo
  const double abw = fabs(bw_mhz), df = abw / nchan;
  const double nu0 = (bw_mhz > 0 ? freq_mhz - abw / 2.0 + 0.5 * df                       // k = 0
                                 : freq_mhz + abw / 2.0 - ((double)(nchan - 1) + 0.5) * df);  // k = C-1
  if (!(nu0 - df / 2.0 > 0)) return "band reaches 0 MHz: cannot dedisperse";
  const double d = fabs(dm) / kDmDispersion;
  const double lo = nu0 - df / 2.0, hi = nu0 + df / 2.0;
  const double t_lo = d * (1.0 / (lo * lo) - 1.0 / (nu0 * nu0));
  const double t_hi = d * (1.0 / (nu0 * nu0) - 1.0 / (hi * hi));
  const double rate = df * 1.0e6;
  const double fpos = ceil(t_hi * rate * 1.05), fneg = ceil(t_lo * rate * 1.05);
  if (!(fpos + fneg < 1.0e6)) return "DM smears more than freq_res can hold";
  int pos = (int)fpos, neg = (int)fneg;
  uint64_t r = freq_res ? freq_res : (nchan <= 128 ? 512u : 2u * nchan);
  while (r < 4ull * (uint64_t)(pos + neg) || r < 2ull * tscrunch) r *= 2;
  if (r > kMaxCoherentFreqRes) {
    std::ostringstream e;
    e << "DM " << dm << " smears " << (pos + neg) << " samples of a " << df << " MHz channel: freq_res would exceed "
      << kMaxCoherentFreqRes;
    return e.str();
  }
  neg += (int)((r - pos - neg) % tscrunch);
  *r_out = (uint32_t)r;
  *nfilt_pos = pos;
  *nfilt_neg = neg;
  return "";
}

std::string make_plan(const frbch_config& cfg, Plan* pl, size_t lds_limit, int in_bits) {
  std::ostringstream e;
  if (!in_bits) in_bits = cfg.input_bits ? (int)cfg.input_bits : 2;
  if (in_bits != 1 && in_bits != 2) return "input_bits must be 1 or 2";
  pl->in_bits = in_bits;
  if (cfg.nchan < 2 || !is_pow2(cfg.nchan) || cfg.nchan > 8192)
    return "nchan must be a power of two in [2, 8192]";
  uint32_t r = cfg.freq_res ? cfg.freq_res : (cfg.nchan <= 128 ? 512u : 2u * cfg.nchan);
  if (r < 2 || !is_pow2(r) || r > 16384) return "freq_res must be a power of two in [2, 16384]";
  const uint32_t t = cfg.tscrunch ? cfg.tscrunch : 1;
  if (!is_pow2(t) || t > r) return "tscrunch must be a power of two not larger than freq_res";
  if (cfg.pol_mode < 0 || cfg.pol_mode > 5) {
    e << "pol = " << cfg.pol_mode << " not implemented. Choices are 0, 1, 2, 3, 4 (and 5 = IQUV)";
    return e.str();
  }
  if (cfg.nbit_out != 2 && cfg.nbit_out != 8 && cfg.nbit_out != 16 && cfg.nbit_out != -32) {
    e << "nbit=" << cfg.nbit_out << " not in supported values of [2, 8, 16, -32]. ";
    return e.str();
  }
  if (cfg.bw_mhz == 0.0 || !(fabs(cfg.bw_mhz) < 1.0e4)) return "BW must be non-zero";
  if (cfg.start_s < 0 || cfg.total_s < 0) return "-S and -T must be non-negative";
  pl->coherent = cfg.coherent ? 1 : 0;
  pl->nfilt_pos = pl->nfilt_neg = 0;
  if (cfg.coherent) {   // -D <dm> -F C:D (process_vdif.py:177-180): R follows from the smearing
    const std::string why = coherent_geometry(cfg.freq_mhz, cfg.bw_mhz, cfg.nchan, cfg.freq_res, t, cfg.dm, &r,
                                              &pl->nfilt_pos, &pl->nfilt_neg);
    if (!why.empty()) return why;
  }

  pl->c = (int)cfg.nchan;
  pl->r = (int)r;
  pl->c2 = 2 * pl->c;
  pl->log2_c2 = ilog2(pl->c2);
  pl->log2_r = ilog2(r);
  pl->n = (uint64_t)pl->c2 * r;
  pl->log2_n = pl->log2_c2 + pl->log2_r;
  pl->log2_nlo = (pl->log2_n + 1) / 2;
  pl->tscr = (int)t;
  pl->nif = cfg.pol_mode >= 4 ? 4 : 1;   // 4: coherency products, 5: Stokes I,Q,U,V
  pl->flip = cfg.bw_mhz > 0 ? 1 : 0;
  pl->nthreads = 256;
  pl->ncol = (uint64_t)pl->nif * pl->c;
  if (pl->ncol % 4) return "nif*nchan must be divisible by 4";
  pl->block_payload_bytes = pl->n * (uint64_t)in_bits / 4;   // 2 pols x in_bits per time sample
  pl->keep = (int)r - pl->nfilt_pos - pl->nfilt_neg;
  pl->hop = (uint64_t)pl->c2 * (uint64_t)pl->keep;
  pl->block_stride_bytes = pl->hop * (uint64_t)in_bits / 4;
  pl->rows_per_block = (uint64_t)pl->keep / t;
  pl->dls_lg_ns = 0;
  if (cfg.unpack_mode > 1) return "unpack_mode must be 0 (static level table) or 1 (dynamic level setting)";
  if (cfg.unpack_mode == 1) {
    const uint32_t ns = cfg.dls_nsample ? cfg.dls_nsample : 512u;
    if (in_bits != 2) return "dynamic level setting needs 2-bit input (a 1-bit sample has no magnitude)";
    if (!is_pow2(ns) || ns < 16 || ns > 8192) return "dls_nsample must be a power of two in [16, 8192]";
    if (pl->n % ns || pl->hop % ns) return "dls_nsample must divide the block length and the distance between block starts";
    if (cfg.dls_threshold < 0.f || cfg.dls_threshold > 4.f) return "dls_threshold must lie in (0, 4] (0 = 0.9674)";
    pl->dls_lg_ns = ilog2(ns);
  }
  pl->k3_lds = 2 * (size_t)(r + 1) * 8;
  pl->k4_lds = 64 * 65 * 4;
  if (pl->coherent && pl->k3_lds > lds_limit) return "freq_res too large for the LDS of this device";
  const int nb = cfg.nbit_out < 0 ? 32 : cfg.nbit_out;
  pl->row_bytes = pl->ncol * nb / 8;

  // LDS tiling: prefer two resident workgroups per CU (80 KiB each), allow one big one.
  const size_t pref = std::min<size_t>(lds_limit, 80 * 1024);
  const size_t seq1 = (size_t)(r + 1) * 8;
  size_t gmax = pref / seq1;
  if (gmax < 1) gmax = 1;
  pl->g = std::min<int>(pow2_floor(gmax), pl->c2);
  pl->k1_lds = (size_t)pl->g * seq1;
  if (pl->k1_lds > lds_limit) return "freq_res too large for the LDS of this device";

  const size_t row1 = (size_t)(pl->c2 + 1) * 8;
  size_t tmax = pref / row1;
  if (tmax < 1) tmax = 1;
  pl->tt = std::min<int>(pow2_floor(tmax), (int)r);
  pl->k2_lds = (size_t)pl->tt * row1;
  if (pl->tscr > pl->tt) pl->k2_lds += pl->ncol * 4;
  if (pl->k2_lds > lds_limit) return "nchan (x tscrunch accumulators) too large for the LDS";
  pl->kc_lds = (size_t)pl->c2 * 8;

  pl->rate_in = 2.0e6 * fabs(cfg.bw_mhz);
  pl->rate_out = fabs(cfg.bw_mhz) * 1.0e6 / ((double)pl->c * t);
  pl->tsamp_s = (double)pl->c * t / (fabs(cfg.bw_mhz) * 1.0e6);  // create_config.py:561
  pl->interval_rows = 0;
  if (cfg.rescale_interval_s > 0) {
    pl->interval_rows = (uint64_t)(cfg.rescale_interval_s * pl->rate_out);  // uint64 cast
    if (pl->interval_rows < 1) pl->interval_rows = 1;
  }
  const double df = fabs(cfg.bw_mhz) / pl->c;
  pl->fch1 = cfg.freq_mhz + fabs(cfg.bw_mhz) / 2.0 - df / 2.0;
  pl->foff = -df;

  switch (cfg.nbit_out) {
    case 2: pl->digi_mean = 1.5f; pl->digi_scale = 1.0f; pl->digi_max = 3.0f; break;
    case 8: pl->digi_mean = 127.5f; pl->digi_scale = (float)(127.5 / 6.0); pl->digi_max = 255.0f; break;
    case 16: pl->digi_mean = 32767.5f; pl->digi_scale = (float)(32767.5 / 6.0); pl->digi_max = 65535.0f; break;
    default: pl->digi_mean = 0.0f; pl->digi_scale = 1.0f; pl->digi_max = 0.0f; break;
  }

  // fast path (kernels_fast.inc): lengths 512..4096 = 256*M, 16 points per (virtual) thread.
  // wave-private variant for M <= 8 (flags & 8 selects the barrier variant instead).
  pl->fast_k1_log2m = pl->fast_k2_log2m = 0;
  pl->fast_k1_wave = pl->fast_k2_wave = 0;
  pl->fast_k2_m1 = 0;
  pl->k2_two_stage = 0;
  pl->k2_stage1_tscr = 0;
  pl->spill_tile_major = 0;
  pl->fast_k2_nt = (cfg.flags & 4u) ? 1024 : 512;
  pl->k1_fast_lds = pl->k2_fast_lds = 0;
  const bool want_wave = !(cfg.flags & 8u);
  if (!(cfg.flags & 1u) && r >= 512 && r <= 8192 && in_bits == 2 && !pl->coherent) {   // the fast gather is written for 2-bit input
    const int m = (int)r / 256;
    const bool wave = want_wave;               // (every M <= 32 has a wave-private K1; flags & 8 asks for the barrier kernels)
    const int tps = 16 * m;
    const int gfast = wave ? (m == 32 ? 2 : (m == 16 ? 4 : 8 * (tps < 64 ? 64 / tps : 1))) : 64 / m;
    const size_t seq = (size_t)r + r / 8 + 8;
    // M = 8 experiments (flags & 64): two half-size workgroups per CU (4 branches each) filling interleaved
    // halves of the 8-branch layout rows, with 2 waves (or, flags & 128, 1 wave) per sequence
    int kind = 0, kg = gfast;
    if (wave && m == 8 && (cfg.flags & 64u)) {   // experiments only: the single 8-branch workgroup measured fastest
      kind = (cfg.flags & 128u) ? 1 : 2;
      kg = 4;
    } else if (wave && m == 8 && (cfg.flags & 128u)) {
      kind = 3;                                   // 16 waves, two per sequence, 8 branches
    } else if (wave && m == 16) {
      kind = 4;                                   // R = 4096: 8 waves, two per sequence, 4 branches
    } else if (wave && m == 32) {
      kind = 5;                                   // R = 8192: 8 waves, four per sequence (two virtual threads per lane), 2 branches
    }
    const size_t lds = (size_t)kg * seq * 8 + (size_t)r * (kg / 2) + 128 + (wave ? (size_t)kg * 132 : 0) +   // + unpack LUT + coarse delay factors + arrival counters
                       ((m >= 16 && !wave) || m == 32 ? (size_t)m * 128 : 0);                               // + the radix-M pass's twiddles (barrier kernels, M >= 16; every kernel at M = 32)
    const size_t generic_lds = (size_t)gfast * seq1;   // fallback for unaligned calls keeps the layout
#ifdef FRBCH_EXPERIMENTS
    static const int gl_env = getenv("FRBCH_GL") ? atoi(getenv("FRBCH_GL")) : 0;   // layout group = workgroup group
#else
    const int gl_env = 0;
#endif
    if (gfast <= pl->c2 && lds <= lds_limit && generic_lds <= lds_limit) {
      pl->fast_k1_log2m = ilog2(m);
      pl->fast_k1_wave = wave ? 1 : 0;
      pl->fast_k1_kind = kind;
      pl->fast_k1_g = kg;
      pl->g = (gl_env && gl_env == kg) ? kg : gfast;
      pl->k1_lds = (size_t)pl->g * seq1;
      pl->k1_fast_lds = lds;
    }
  }
  if (!(cfg.flags & 2u) && pl->c2 >= 256 && pl->c2 <= 8192 && pl->g >= 2 && !pl->coherent) {
    const int m = pl->c2 / 256;
    // 2C = 8192 (M = 32): the wave kernel takes TWO time samples per workgroup (four waves and two virtual threads per lane
    // each; persistent, the next tile prefetched into registers); one product and tscrunch <= 2 only -- everything else
    // stays on the barrier kernels
    // tscrunch > 2: two stages -- the same kernel with two time samples per row into a scratch buffer, then frbch_k2_scrunch
    // (K2 1.98 + 0.2 ms instead of the barrier kernel walking its sub-tiles, 3.85 ms at -t 8)
    const bool wave32 = m == 32 && pl->nif == 1 && pl->fast_k1_log2m == 5 && pl->g == 2 && !(cfg.flags & (1u << 21)) &&
                        (pl->tscr <= 2 || (pl->tscr % 2 == 0 && pl->tscr <= (int)r));
    pl->k2_two_stage = (wave32 && want_wave && !(cfg.flags & 4u) && pl->tscr > 2) ? pl->tscr / 2 : 0;
    pl->k2_stage1_tscr = 2;
    const bool wave = want_wave && (m <= 16 || wave32) && !(cfg.flags & 4u);
    const int tps = 16 * m;
    const int spw = tps < 64 ? 64 / tps : 1;
    // the same beyond the largest tile of the other wave K2s (8 sequences x spw; 4 at 2C = 4096): -t 16 at 1024 channels
    // ran on the generic K2
    {
      const int tmax = m == 16 ? 4 : 8 * spw;
      if (wave && m <= 16 && pl->tscr > tmax && pl->tscr % tmax == 0 && pl->tscr <= (int)r) {
        pl->k2_two_stage = pl->tscr / tmax;
        pl->k2_stage1_tscr = tmax;
      }
    }
    // wave variant: 2 waves per workgroup (more, smaller workgroups resident per CU) when tscrunch
    // allows it and flags & 16 does not ask for 4
    // sequences (time samples) per workgroup = nw*spw: the smallest of 2, 4, 8 (x spw) that holds one
    // tscrunch group; small workgroups keep more of them resident per CU
    int nw = 4;
    if (wave && m == 16)
      nw = 4;                                     // 2C = 4096: 4 time samples per workgroup (128-B gather runs), two waves each
    else if (wave && !(cfg.flags & 16u) && pl->tscr <= 2 * spw && ((2 * spw * pl->g) % 2 == 0) &&
        (size_t)2 * spw * pl->ncol * 4 <= (size_t)2 * spw * ((size_t)pl->c2 + pl->c2 / 8 + 8) * 8)
      nw = 2;
    else if (wave && pl->tscr > 4 * spw)
      nw = 8;
    pl->fast_k2_nw = nw;
    // 2C = 8192: one workgroup = 2 time samples (147 KB of LDS, one workgroup per CU), or -- no tscrunch -- ONE time
    // sample with 512 threads (74 KB: two workgroups per CU, one gathers while the other transforms)
    if (m == 32) pl->fast_k2_nt = (pl->tscr == 1 && !(cfg.flags & 4u)) ? 512 : 1024;
    const int tt = wave ? (m == 32 ? 2 : nw * spw) : pl->fast_k2_nt / tps;
    // 2C = 8192: a workgroup may walk tscrunch/tt tiles and add them up in registers
    const bool walk = m == 32 && !wave && pl->nif == 1 && pl->tscr > tt && pl->tscr % tt == 0 && pl->tscr <= (int)r;
    const size_t seq = (size_t)pl->c2 + pl->c2 / 8 + ((m == 32 && !wave) ? 0 : 8);
    const size_t lds = (size_t)tt * seq * 8 + (walk ? (size_t)pl->c * 4 : 0) +   // + the row of sub-tile sums
                       ((m == 32 && (pl->fast_k2_nt == 512 || wave)) ? 4096 : 0);   // + the radix-32 pass's twiddles (single-sample K2)
    // 2C = 8192, four products, tscrunch > 2 (the IQUV spelling of config 4's `-t 8`): the barrier K2 writes rows of its
    // two-sample tile into the scratch buffer and frbch_k2_scrunch adds them up (was: the generic K2, 3x slower)
    bool barrier_two_stage = false;
    if (m == 32 && !wave && pl->nif == 4 && pl->tscr > 2 && pl->tscr % 2 == 0 && pl->tscr <= (int)r && pl->fast_k1_log2m == 5) {
      pl->k2_two_stage = pl->tscr / 2;
      pl->k2_stage1_tscr = 2;
      barrier_two_stage = true;
    }
    if (tt >= 1 && (pl->tscr <= tt || walk || pl->k2_two_stage) && tt <= (int)r && (size_t)tt * pl->ncol * 4 <= lds && lds <= lds_limit &&
        (tt * pl->g) % 2 == 0 && (m > 1 || wave)) {   // 2C = 256 (radix 16 x 16): wave-private kernel only
      pl->fast_k2_log2m = ilog2(m);
      pl->fast_k2_m1 = m == 1;
      pl->fast_k2_wave = wave ? 1 : 0;
      pl->k2_fast_lds = lds;
    }
    if ((!pl->fast_k2_wave && !barrier_two_stage) || !(pl->fast_k2_log2m || pl->fast_k2_m1)) pl->k2_two_stage = 0;
  }

  // few channels (2C = 64 / 128: the 32 / 64 channels per IF of the online chain, submit_job.py:74-105): one lane (pair)
  // keeps a whole across-branch sequence in registers (frbch_k2_lane); reads the slab layout of 16-branch groups the
  // R = 512 wave K1 (or the generic K1 in its place) writes
  pl->fast_k2_lane = 0;
  if (!(cfg.flags & 2u) && !pl->coherent && (pl->c2 == 64 || pl->c2 == 128) && pl->g == 16 && !pl->fast_k2_log2m && !pl->fast_k2_m1) {
    const int nh = pl->c2 / 64;
    if (pl->tscr <= 64 / nh && ((int)r * nh) % 256 == 0) pl->fast_k2_lane = nh;
  }

  // R = 2048, flag bit 24: the split K1 (bin-parity halves, 16 independent waves per CU) instead of the paired-branch wave
  // K1 whenever frbch_k0_stage has corner-turned the batch (measured slower: 1.76 vs 1.54 ms)
  pl->fast_k1_split = 0;
  pl->k1_split_lds = (size_t)2 * 8 * (1024 + 128 + 2) * 8 + 2048 * 4 + (16 + 8 * 16 + 32) * 8;
  if (pl->fast_k1_wave && pl->fast_k1_log2m == 3 && pl->fast_k1_kind == 0 && pl->fast_k1_g == 8 && pl->g == 8 &&
      (cfg.flags & (1u << 24)) && pl->k1_split_lds <= lds_limit)
    pl->fast_k1_split = 1;

  // tile-major spill: the paired-branch wave K1 (R = 2048, 8 branches per workgroup) in front of a wave K2 whose
  // workgroup takes two time samples (measured: K2 1.34 -> 1.29 ms, its gather alone 1.06 -> 0.95 ms; K1 unchanged)
  pl->spill_tile_major = 0;
  if (pl->fast_k1_wave && pl->fast_k1_log2m == 3 && pl->fast_k1_kind == 0 && pl->fast_k1_g == 8 && pl->g == 8 &&
      pl->fast_k2_wave && !pl->coherent && !(cfg.flags & (1u << 21))) {   // flag bit 21: keep the slab layout (A/B tests)
    const int m2 = pl->c2 / 256, tps2 = 16 * m2, spw2 = tps2 < 64 ? 64 / tps2 : 1;
    const bool two = (m2 == 16) || (m2 == 8 && !(cfg.flags & 32u) && pl->fast_k2_nw != 8);   // two waves per sequence
    const int tt2 = two ? (pl->fast_k2_nw == 2 ? 2 : 4) : pl->fast_k2_nw * spw2;
    if (tt2 == 2) pl->spill_tile_major = 2;
  }
  // 2C = 2048 behind that K1: the wave-private K2 (one wave per time sample, four per workgroup, two workgroups per CU; tscrunch up
  // to its four-sample tile).  flag bit 26 keeps frbch_k2_wave (kernel-family cross-checks)
  pl->fast_k2_priv = 0;
  pl->k2_priv_lds = (size_t)4 * (2048 + 256) * 8 + 1024 * 8;      // = 81,920: exactly half of the CU's 160 KiB
  if (pl->fast_k1_wave && pl->fast_k1_log2m == 3 && pl->fast_k1_kind == 0 && pl->fast_k1_g == 8 && pl->g == 8 && !pl->fast_k1_split &&
      pl->fast_k2_wave && pl->fast_k2_log2m == 3 && pl->c2 == 2048 && !pl->coherent && !(cfg.flags & ((1u << 21) | (1u << 26))) &&
      pl->k2_priv_lds <= lds_limit && pl->tscr <= 4 && !pl->k2_two_stage) {
    pl->fast_k2_priv = 1;
    pl->spill_tile_major = 2;      // (what the paired-branch K1 writes for it, whatever tile the two-wave K2 would have taken)
  }
  // chunks of eight time samples between the M = 32 barrier kernels (2 branches per K1 workgroup: the slab layout
  // leaves K2 one 32-byte piece per 128-KB slab)
  if (pl->fast_k1_log2m == 5 && pl->fast_k2_log2m == 5 && !pl->coherent && !(cfg.flags & (1u << 21)))
    pl->spill_tile_major = 8;   // (the wave K2 at 2C = 8192 reads this layout only: its plan condition above repeats these terms)

  // coherent pipeline on the register-pass kernels (barrier variants): K1 forward-only + K3 need R = 256*M, K2c needs
  // 2C = 256*M', M, M' in 2..32; the K1 group (64/M branches) becomes the layout group of the first spill
  pl->coh_fast_r = pl->coh_fast_c = 0;
  pl->k2c_fast_lds = pl->k3_fast_lds = 0;
  // R = 4096 (M = 16, config 5): 512-thread K3 workgroups (one channel and its mirror), two per CU: one gathers while
  // the other transforms (K3 2.55 -> 2.20 ms)
  pl->coh_nt = (pl->coherent && r == 4096 && !(cfg.flags & 4u)) ? 512 : 1024;
  if (pl->coherent && !(cfg.flags & 1u) && r >= 512 && r <= 8192 && in_bits == 2 && pl->c >= 4) {
    const int m = (int)r / 256;
    const int gfast = 64 / m;
    const size_t seq = (size_t)r + r / 8 + 8;
    const size_t lds1 = (size_t)gfast * seq * 8 + (size_t)r * (gfast / 2 ? gfast / 2 : 1) + (m >= 16 ? (size_t)m * 128 : 0);
    const size_t seq3 = (size_t)r + r / 8 + (m == 32 ? 0 : 8);
    const int ns = pl->coh_nt / (16 * m);           // sequences per K3 workgroup (pairs of rows)
    const size_t lds3 = (size_t)ns * seq3 * 8 + (m >= 16 ? (size_t)m * 128 : 0);   // + the radix-M pass's twiddles
    if (gfast >= 2 && gfast <= pl->c2 && lds1 <= lds_limit && lds3 <= lds_limit && ns >= 2 && pl->c % (ns / 2) == 0 &&
        (size_t)gfast * seq1 <= lds_limit) {
      pl->coh_fast_r = ilog2(m);
      pl->fast_k1_log2m = ilog2(m);
      pl->fast_k1_wave = 0;
      pl->fast_k1_kind = 0;
      pl->fast_k1_g = gfast;
      pl->g = gfast;
      pl->k1_lds = (size_t)gfast * seq1;
      pl->k1_fast_lds = lds1;
      pl->k3_fast_lds = lds3;
      // R = 4096 (config 5): the wave K1 (8 waves, two per branch, 4 branches = the same layout group) in its forward-only form
      const size_t ldsw = (size_t)gfast * seq * 8 + (size_t)r * (gfast / 2) + 128 + (size_t)gfast * 132;
      if (m == 16 && want_wave && ldsw <= lds_limit) {
        pl->fast_k1_wave = 1;
        pl->fast_k1_kind = 4;
        pl->k1_fast_lds = ldsw;
      }
    }
  }
  if (pl->coherent && !(cfg.flags & 2u) && pl->c2 >= 512 && pl->c2 <= 8192 && pl->g >= 2) {
    const int m = pl->c2 / 256;
    const size_t seq = (size_t)pl->c2 + pl->c2 / 8 + (m == 32 ? 0 : 8);
    const int tt = 1024 / (16 * m);   // (K2c stays at 1024 threads: with 512 its gather pieces and store runs halve, 2.81 -> 3.6 ms)
    const size_t lds = (size_t)tt * seq * 8 + (m >= 16 ? (size_t)m * 128 : 0);   // + the radix-M pass's twiddles
    if (tt >= 1 && tt <= (int)r && (int)r % tt == 0 && lds <= lds_limit && (tt * pl->g) % 2 == 0 && ((size_t)tt * pl->c2 / 2) % 1024 == 0) {
      pl->coh_fast_c = ilog2(m);
      pl->k2c_fast_lds = lds;
    }
  }

  uint32_t maxb = cfg.max_blocks_per_launch;
  if (!maxb) {
    // big batches amortise launches and let the persistent kernels run many iterations: up to 8 GiB of spill, 5 GiB with four products (the power
    // buffer, the staging areas and a scan's row buffer grow with the batch and with the products; the eight handles of a
    // config-3 scan take ~12 GB each: the card has 288).  Callers split their blocks
    // into EQUAL batches (measured, 152-block scan: 64 + 64 + 24 blocks 3.60 ms per step, 3 x 51 3.61 ms, 76 + 76
    // 3.48 ms, one batch of 152 3.40-3.47 ms).
    const uint64_t spill_per_block = pl->n * 8 * (pl->coherent ? 2 : 1);
    // (four products: one batch of 152 blocks instead of 2 x 76 measured the same on its own, 130.1 vs 130.5 Gsamples/s, but the
    // digitiser of IF i beside the K1 of IF i + 1 wants ONE long K1 launch: 37.25 -> 36.4 ms per 8-IF step)
    // (the spill is allocated for maxb blocks: 5 GiB with four products = 160 blocks of 2^22 samples, the 152 of a 10-s IF in one piece;
    // 8 GiB would double what the eight handles of a config-3 scan hold for nothing)
    maxb = (uint32_t)std::max<uint64_t>(1, ((cfg.pol_mode >= 4 ? 5120ull : 8192ull) << 20) / spill_per_block);
    // at most 256 blocks per launch, or 2^28 samples of small blocks (32 .. 128 channels: a block is 2^15 .. 2^17 samples and 10 s of
    // an IF thousands of blocks -- 256 per launch made every kernel launch-bound: 39 launches of ~13 us for 0.5 ms of Kc)
    const uint64_t cap = std::max<uint64_t>(256, (1ull << 28) / pl->n);
    if (maxb > cap) maxb = (uint32_t)cap;
  }
  if (maxb > 32768) maxb = 32768;
  while (maxb > 1 && (uint64_t)maxb * pl->block_payload_bytes > (1ull << 31)) maxb /= 2;  // 32-bit launch-relative offsets
  pl->maxb = maxb;
  // Spill slabs of one block are R*g*8 B = a power of two apart: the 2C/g pieces K2 gathers for one tile would
  // all sit on one HBM channel.  A pad per slab spreads them.
#ifdef FRBCH_EXPERIMENTS
  static const int pad_env = getenv("FRBCH_SPILL_PAD") ? atoi(getenv("FRBCH_SPILL_PAD")) : -1;   // in cf
#else
  const int pad_env = -1;
#endif
  // measured (K2, cfg 2): pad 0 -> 1.55 ms, 16 -> 1.38, 48..8208 -> 1.29-1.33; 272 cf = 2 KB + 128 B
  const int pad = (uint64_t)pl->r * pl->g >= 2048 ? 272 : 16;
  pl->gs = (uint64_t)pl->r * pl->g + (uint64_t)(pad_env >= 0 ? pad_env : pad);
  return "";
}

// ---------------------------------------------------------------------------------------------
// Dynamic level setting (frbch_config.unpack_mode 1): the table the unpack looks its two output levels up in.
// A Gaussian voltage of rms s sampled with threshold t falls inside the threshold with probability Phi = erf(t / (s sqrt 2)); the
// mean power of the samples inside is s^2 (1 - g / Phi), of those outside s^2 (1 + g / (1 - Phi)), g = sqrt(2 / pi) u exp(-u^2 / 2),
// u = t / s.  A window with k of L samples inside estimates Phi = k / L, hence u = sqrt 2 erfinv(Phi) and s = t / u (t in units of the
// nominal rms: cfg.dls_threshold); its output levels are the roots of those two mean powers, so that Phi lo^2 + (1 - Phi) hi^2 = s^2:
// the unpacked power follows the undigitised power (Jenet & Anderson 1998, PASP 110, 1467: dynamic level setting with
// power-conserving output levels; the formulas are this build's restatement, DESIGN.md section 2a -- DSPSR is absent).
// Windows whose count is further than `cutoff` standard deviations sqrt(L Phi0 (1 - Phi0)) from L Phi0, Phi0 = erf(t / sqrt 2), and the
// counts 0 and L (no estimate) get (0, 0): zeroed.  (The test oracle restates this table; tests/test_dynamic_levels.py compares.)
// ---------------------------------------------------------------------------------------------
static double erfinv_unit(double y) {   // y in (0, 1)
  const double a = 0.147, l = log((1.0 - y) * (1.0 + y)), b = 2.0 / (M_PI * a) + 0.5 * l;
  double x = sqrt(sqrt(b * b - l / a) - b);   // (Winitzki's approximation: the start of the Newton iteration)
  for (int it = 0; it < 80; ++it) {
    const double err = erf(x) - y;
    const double step = err / (2.0 / sqrt(M_PI) * exp(-x * x));
    x -= step;
    if (fabs(step) <= 1e-16 * fabs(x)) break;
  }
  return x;
}
std::vector<float> dls_table(uint32_t nsample, float cutoff_sigma, float threshold) {
  const double thr = threshold > 0.f ? (double)threshold : 0.9674;
  const double cutoff = cutoff_sigma == 0.f ? 10.0 : (double)cutoff_sigma;
  const double phi0 = erf(thr / sqrt(2.0)), mean = nsample * phi0, sd = sqrt(nsample * phi0 * (1.0 - phi0));
  long kmin = 1, kmax = (long)nsample - 1;
  if (cutoff > 0) {
    kmin = std::max<long>(kmin, (long)ceil(mean - cutoff * sd));
    kmax = std::min<long>(kmax, (long)floor(mean + cutoff * sd));
  }
  std::vector<float> tab(2 * ((size_t)nsample + 1), 0.f);
  for (long k = kmin; k <= kmax; ++k) {
    const double phi = (double)k / (double)nsample;
    const double u = sqrt(2.0) * erfinv_unit(phi);
    const double sig = thr / u;
    const double g = sqrt(2.0 / M_PI) * u * exp(-0.5 * u * u);
    tab[2 * k] = (float)(sig * sqrt(1.0 - g / phi));
    tab[2 * k + 1] = (float)(sig * sqrt(1.0 + g / (1.0 - phi)));
  }
  return tab;
}

void fill_twiddles(float* dst, uint64_t n, uint64_t count, uint64_t step) {
  for (uint64_t k = 0; k < count; ++k) {
    const uint64_t q = (k * step) % n;
    // reduce to the first octant-ish range through exact symmetries for accuracy
    const double a = -2.0 * M_PI * (double)q / (double)n;
    dst[2 * k] = (float)cos(a);
    dst[2 * k + 1] = (float)sin(a);
  }
}

// ---------------------------------------------------------------------------------------------
// SIGPROC header (public sigproc format; consumer is `splice`, base2fil.sh:422)
// ---------------------------------------------------------------------------------------------
static void put_str(std::vector<uint8_t>& v, const std::string& s) {
  const int32_t n = (int32_t)s.size();
  const uint8_t* p = (const uint8_t*)&n;
  v.insert(v.end(), p, p + 4);
  v.insert(v.end(), s.begin(), s.end());
}
static void put_i(std::vector<uint8_t>& v, const char* key, int32_t x) {
  put_str(v, key);
  const uint8_t* p = (const uint8_t*)&x;
  v.insert(v.end(), p, p + 4);
}
static void put_d(std::vector<uint8_t>& v, const char* key, double x) {
  put_str(v, key);
  const uint8_t* p = (const uint8_t*)&x;
  v.insert(v.end(), p, p + 8);
}

double sigproc_angle(const char* text) {
  std::string t(text);
  size_t b = t.find_first_not_of(" \t");
  if (b == std::string::npos) return 0.0;
  t = t.substr(b);
  double sign = 1.0;
  if (t[0] == '-') sign = -1.0;
  while (!t.empty() && (t[0] == '-' || t[0] == '+')) t.erase(0, 1);
  double part[3] = {0, 0, 0};
  std::stringstream ss(t);
  std::string item;
  for (int i = 0; i < 3 && std::getline(ss, item, ':'); ++i) part[i] = atof(item.c_str());
  return sign * (part[0] * 10000.0 + part[1] * 100.0 + part[2]);
}

int sigproc_telescope_id(const char* name) {
  static const struct { const char* n; int id; } tab[] = {
      {"fake", 0}, {"arecibo", 1}, {"ooty", 2}, {"nancay", 3}, {"parkes", 4}, {"jodrell", 5},
      {"gbt", 6}, {"gmrt", 7}, {"effelsberg", 8}, {"ata", 9}, {"srt", 10}, {"lofar", 11},
      {"vla", 12}};
  std::string s(name);
  std::transform(s.begin(), s.end(), s.begin(), ::tolower);
  for (auto& t : tab)
    if (s == t.n) return t.id;
  return 0;
}

std::vector<uint8_t> sigproc_header(const frbch_config& cfg, const Plan& pl, double tstart_mjd, int nchans_total) {
  std::vector<uint8_t> v;
  put_str(v, "HEADER_START");
  put_i(v, "telescope_id", sigproc_telescope_id(cfg.telescope));
  put_i(v, "machine_id", 0);
  put_i(v, "data_type", 1);
  std::string raw(cfg.datafile);
  if (raw.size() > 80) raw = raw.substr(raw.size() - 80);
  put_str(v, "rawdatafile");
  put_str(v, raw);
  std::string src(cfg.source);
  if (src.size() > 80) src.resize(80);
  put_str(v, "source_name");
  put_str(v, src);
  put_i(v, "barycentric", 0);
  put_i(v, "pulsarcentric", 0);
  put_d(v, "az_start", 0.0);
  put_d(v, "za_start", 0.0);
  put_d(v, "src_raj", sigproc_angle(cfg.ra));
  put_d(v, "src_dej", sigproc_angle(cfg.dec));
  put_d(v, "tstart", tstart_mjd);
  put_d(v, "tsamp", pl.tsamp_s);
  put_i(v, "nbits", cfg.nbit_out == -32 ? 32 : cfg.nbit_out);
  put_d(v, "fch1", pl.fch1);
  put_d(v, "foff", pl.foff);
  put_i(v, "nchans", nchans_total > 0 ? nchans_total : pl.c);   // > 0: several IFs side by side (frbch_run_scan)
  put_i(v, "nifs", pl.nif);
  put_d(v, "refdm", cfg.dm);
  put_str(v, "HEADER_END");
  return v;
}

}  // namespace frbch

// =============================================================================================
// C ABI: configuration helpers (no GPU needed)
// =============================================================================================
using namespace frbch;

static void set_err(char* err, size_t cap, const std::string& s) {
  if (err && cap) snprintf(err, cap, "%s", s.c_str());
}

extern "C" int frbch_config_init(frbch_config* cfg) {
  if (!cfg) return FRBCH_E_ARG;
  memset(cfg, 0, sizeof(*cfg));
  cfg->size = (uint32_t)sizeof(*cfg);
  cfg->abi_version = FRBCH_ABI_VERSION;
  cfg->freq_mhz = 1608.0;           // process_vdif.py:19
  cfg->bw_mhz = 16.0;               // process_vdif.py:28
  cfg->start_s = 0.0;
  cfg->total_s = 1.0e30;            // digifil without -T: to end of data
  cfg->nchan = 128;                 // run_digifil default (process_vdif.py:142)
  cfg->freq_res = 0;
  cfg->tscrunch = 1;
  cfg->nbit_out = 8;
  cfg->pol_mode = 2;
  cfg->rescale_constant = 0;
  cfg->rescale_interval_s = 10.0;   // digifil's default rescale interval
  cfg->dm = 0.0;
  cfg->coherent = 0;
  cfg->device = 0;
  snprintf(cfg->telescope, sizeof cfg->telescope, "ONSALA85");  // process_vdif.py:36
  snprintf(cfg->source, sizeof cfg->source, "unknown");
  snprintf(cfg->ra, sizeof cfg->ra, "00:00:00.0");
  snprintf(cfg->dec, sizeof cfg->dec, "00:00:00.0");
  return FRBCH_OK;
}

extern "C" int frbch_config_from_hdr(const char* hdr_path, frbch_config* cfg) {
  if (!hdr_path || !cfg) return FRBCH_E_ARG;
  std::ifstream f(hdr_path);
  if (!f) return FRBCH_E_IO;
  std::string line;
  bool have_freq = false, have_bw = false, have_file = false, is_vdif = false;
  while (std::getline(f, line)) {
    std::istringstream ls(line);
    std::string key;
    if (!(ls >> key)) continue;
    std::string val;
    std::getline(ls, val);
    const size_t b = val.find_first_not_of(" \t");
    val = b == std::string::npos ? "" : val.substr(b);
    while (!val.empty() && (val.back() == '\r' || val.back() == ' ' || val.back() == '\t')) val.pop_back();
    // a value that does not fit its field is an argument error, not a silently truncated name
    if ((key == "TELESCOPE" && val.size() >= sizeof cfg->telescope) || (key == "SOURCE" && val.size() >= sizeof cfg->source) ||
        (key == "RA" && val.size() >= sizeof cfg->ra) || (key == "DEC" && val.size() >= sizeof cfg->dec) ||
        (key == "DATAFILE" && val.size() >= sizeof cfg->datafile))
      return FRBCH_E_ARG;
    if (key == "TELESCOPE") snprintf(cfg->telescope, sizeof cfg->telescope, "%s", val.c_str());
    else if (key == "SOURCE") snprintf(cfg->source, sizeof cfg->source, "%s", val.c_str());
    else if (key == "RA") snprintf(cfg->ra, sizeof cfg->ra, "%s", val.c_str());
    else if (key == "DEC") snprintf(cfg->dec, sizeof cfg->dec, "%s", val.c_str());
    else if (key == "FREQ") { cfg->freq_mhz = atof(val.c_str()); have_freq = true; }
    else if (key == "BW") { cfg->bw_mhz = atof(val.c_str()); have_bw = true; }
    else if (key == "DATAFILE") { snprintf(cfg->datafile, sizeof cfg->datafile, "%s", val.c_str()); have_file = true; }
    else if (key == "INSTRUMENT") is_vdif = (val == "VDIF");
    else if (key == "NPOL") { if (atoi(val.c_str()) != 2) return FRBCH_E_FORMAT; }
  }
  if (!have_freq || !have_bw || !have_file || !is_vdif) return FRBCH_E_FORMAT;
  return FRBCH_OK;
}

// value of an option that may be attached ("-b8") or separate ("-t 8")
static bool opt_value(const std::string& tok, size_t optlen, int argc, const char* const* argv, int* i,
                      std::string* val) {
  if (tok.size() > optlen) {
    *val = tok.substr(optlen);
    return true;
  }
  if (*i + 1 >= argc) return false;
  *val = argv[++*i];
  return true;
}

extern "C" int frbch_parse_digifil_argv(int argc, const char* const* argv, frbch_config* cfg,
                                        char* hdr_path, size_t hdr_cap, char* out_path,
                                        size_t out_cap, char* err, size_t err_cap) {
  if (!cfg || !argv) return FRBCH_E_ARG;
  std::string hdr, out, val;
  bool want_iquv = false;
  for (int i = 1; i < argc; ++i) {
    const std::string tok = argv[i];
    if (tok.empty()) continue;
    if (tok[0] != '-' || tok.size() == 1) {
      if (!hdr.empty()) { set_err(err, err_cap, "more than one input file: " + tok); return FRBCH_E_ARG; }
      hdr = tok;
      continue;
    }
    bool ok = true;
    if (tok == "-cont") continue;                       // contiguous input: what we assume anyway
    else if (tok == "-c") cfg->rescale_constant = 1;
    else if (tok == "-2") continue;                     // 2-bit excision off: static level table
    // DSPSR's unpacker options, attached to the -2: any of them selects the dynamic level setting (cfg.unpack_mode 1)
    else if (tok.compare(0, 3, "-2n") == 0 && tok.size() > 3) { cfg->unpack_mode = 1; cfg->dls_nsample = (uint32_t)atoi(tok.c_str() + 3); }
    else if (tok.compare(0, 3, "-2c") == 0 && tok.size() > 3) { cfg->unpack_mode = 1; cfg->dls_cutoff_sigma = (float)atof(tok.c_str() + 3); }
    else if (tok.compare(0, 3, "-2t") == 0 && tok.size() > 3) { cfg->unpack_mode = 1; cfg->dls_threshold = (float)atof(tok.c_str() + 3); }
    else if (tok == "-iquv") want_iquv = true;          // extension: Stokes I,Q,U,V from the -d4 products (pol_mode 5)
    else if (tok == "-threads") { ok = opt_value(tok, 8, argc, argv, &i, &val); }
    else if (tok.compare(0, 2, "-b") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->nbit_out = atoi(val.c_str()); }
    else if (tok.compare(0, 2, "-S") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->start_s = atof(val.c_str()); }
    else if (tok.compare(0, 2, "-T") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->total_s = atof(val.c_str()); }
    else if (tok.compare(0, 2, "-D") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->dm = atof(val.c_str()); }
    else if (tok.compare(0, 2, "-t") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->tscrunch = (uint32_t)atoi(val.c_str()); }
    else if (tok.compare(0, 2, "-o") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &out); }
    else if (tok.compare(0, 2, "-I") == 0) { ok = opt_value(tok, 2, argc, argv, &i, &val); cfg->rescale_interval_s = atof(val.c_str()); }
    else if (tok.compare(0, 2, "-P") == 0) {
      ok = opt_value(tok, 2, argc, argv, &i, &val);
      const int p = atoi(val.c_str());
      if (p != 0 && p != 1) { set_err(err, err_cap, "-P must be 0 or 1"); return FRBCH_E_ARG; }
      cfg->pol_mode = p;
    } else if (tok.compare(0, 2, "-d") == 0) {
      ok = opt_value(tok, 2, argc, argv, &i, &val);
      const int d = atoi(val.c_str());
      if (d == 1) cfg->pol_mode = 2;
      else if (d == 3) cfg->pol_mode = 3;
      else if (d == 4) cfg->pol_mode = 4;
      else { set_err(err, err_cap, "-d" + val + " not implemented (choices: 1, 3, 4)"); return FRBCH_E_ARG; }
    } else if (tok.compare(0, 2, "-F") == 0) {
      ok = opt_value(tok, 2, argc, argv, &i, &val);
      const size_t colon = val.find(':');
      const int nch = atoi(val.substr(0, colon).c_str());
      if (nch < 1) { set_err(err, err_cap, "bad -F " + val); return FRBCH_E_ARG; }
      cfg->nchan = (uint32_t)nch;
      if (colon != std::string::npos) {
        const std::string rest = val.substr(colon + 1);
        if (rest == "D") cfg->coherent = 1;
        else {
          const int fr = atoi(rest.c_str());
          if (fr < 1) { set_err(err, err_cap, "bad -F " + val); return FRBCH_E_ARG; }
          cfg->freq_res = (uint32_t)fr;
          cfg->coherent = 0;
        }
      }
    } else {
      set_err(err, err_cap, "unknown option " + tok);
      return FRBCH_E_ARG;
    }
    if (!ok) { set_err(err, err_cap, "option " + tok + " needs a value"); return FRBCH_E_ARG; }
  }
  if (want_iquv) {
    if (cfg->pol_mode != 4) { set_err(err, err_cap, "-iquv needs -d4"); return FRBCH_E_ARG; }
    cfg->pol_mode = 5;
  }
  if (hdr.empty()) { set_err(err, err_cap, "no input .hdr given"); return FRBCH_E_ARG; }
  if (out.empty()) { set_err(err, err_cap, "no output file (-o) given"); return FRBCH_E_ARG; }
  if (hdr_path && hdr_cap) snprintf(hdr_path, hdr_cap, "%s", hdr.c_str());
  if (out_path && out_cap) snprintf(out_path, out_cap, "%s", out.c_str());
  return FRBCH_OK;
}

extern "C" const char* frbch_strerror(int code) {
  switch (code) {
    case FRBCH_OK: return "ok";
    case FRBCH_E_ARG: return "bad argument or unsupported configuration";
    case FRBCH_E_IO: return "I/O error";
    case FRBCH_E_FORMAT: return "unsupported or corrupt input format";
    case FRBCH_E_DEVICE: return "GPU not available or HIP error (there is no CPU fallback)";
    case FRBCH_E_NOMEM: return "out of memory";
    case FRBCH_E_STATE: return "call sequence error";
    case FRBCH_E_CAPACITY: return "output buffer too small";
    default: return "unknown error";
  }
}
