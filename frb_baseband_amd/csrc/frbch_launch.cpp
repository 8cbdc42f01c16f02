// libfrbch: plan -> kernel launches.  The only translation unit that includes the kernel sources: every HIP kernel of the library is
// instantiated and launched here (frbch_internal.h lists the units).
#include "frbch_internal.h"

#include "kernels_generic.inc"
#ifndef FRBCH_NO_FAST
#include "kernels_fast.inc"
#include "kernels_k2priv.inc"
#endif

namespace frbchi {

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
  const size_t lds_msk = pl.k1_fast_lds + (size_t)(LOG2M >= 5 ? pl.r / 8 : pl.r);   // + one flag per n2 row (frames flagged invalid / fillers: MSK): a byte, at R = 8192 a bit
#define FRBCH_K1W(L, NWV, WPSV, NTV)                                                                                       \
  do {                                                                                                                  \
    if (p.stg && p.fbad) hipLaunchKernelGGL((fast::frbch_k1_wave<L, NWV, WPSV, true, false, true>), dim3(ngrp, ny), dim3(NTV), lds_msk, s, p);  \
    else if (p.stg) hipLaunchKernelGGL((fast::frbch_k1_wave<L, NWV, WPSV, true>), dim3(ngrp, ny), dim3(NTV), pl.k1_fast_lds, s, p);  \
    else hipLaunchKernelGGL((fast::frbch_k1_wave<L, NWV, WPSV, false>), dim3(ngrp, ny), dim3(NTV), pl.k1_fast_lds, s, p);        \
  } while (0)
  if constexpr (LOG2M == 5) {
    FRBCH_K1W(5, 8, 4, 512);      // R = 8192: two branches per workgroup, four waves (two virtual threads per lane) each
  } else if constexpr (LOG2M == 4) {
    if (p.coherent) {   // forward transform + delay only, spectrum spilled (K2c follows)
      if (p.stg && p.fbad) hipLaunchKernelGGL((fast::frbch_k1_wave<4, 8, 2, true, true, true>), dim3(ngrp, ny), dim3(512), lds_msk, s, p);
      else if (p.stg) hipLaunchKernelGGL((fast::frbch_k1_wave<4, 8, 2, true, true>), dim3(ngrp, ny), dim3(512), pl.k1_fast_lds, s, p);
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
// which K2 a launch of a plan with frbch_k2_priv takes: float rows stay on frbch_k2_wave (measured: four products, config 3, 1.87 ms
// per IF against 1.97; one product, config 2, 1.15 - 1.20 against 1.20 - 1.28 -- the two-wave kernel reads whole 128-byte lines,
// frbch_k2_priv halves of them twice, and writing float rows leaves less of the memory pipe to hide that), codes and statistics-only
// passes run frbch_k2_priv (steady state of config 3 + 2.4 %, config 2 + 4 %)
bool pol_mode_no_sums(int pol_mode) { return pol_mode == 3; }   // (PP+QQ)^2: its square overflows the fp32 partial sums (~1e24 squared)
bool priv_takes(const Plan& pl, const KParams& p, int priv_grid) {
  // (the two-wave kernel reads the tile-major spill only in its two-sample form: fast_k2_nw == 2, tscrunch <= 2)
  return pl.fast_k2_priv && p.tile_major == 2 && priv_grid > 0 &&
         !(p.out_mode == FRBCH_OUT_FLOAT_POWER && pl.fast_k2_nw == 2 && pl.fast_k2_log2m == 3);
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
    if (p.fbad) {   // flagged frames: the staged wave K1 masks them (a flag per row beside the stage: a byte, at R = 8192 a bit)
      if (!q.stg || pl.fast_k1_kind == 1 || pl.fast_k1_kind == 2 || pl.fast_k1_kind == 3 || pl.fast_k1_split ||
          pl.k1_fast_lds + (size_t)(pl.fast_k1_log2m >= 5 ? pl.r / 8 : pl.r) > h->lds_limit)
        return false;
      q.fbad_frame0 = p.fbad_frame0 + fr0;
    }
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
  // a workgroup keeps its positions (and their kernel factors, in registers) and walks the blocks of the launch; enough workgroups
  // for a few rounds per CU, else the block index strides too
  p.nblk = nb;
  const uint32_t ntile = (uint32_t)(pl.r / tt);
  const uint32_t want = 4u * (uint32_t)std::max(1, h->lane_ncu);
  const uint32_t gy = std::max<uint32_t>(1, std::min<uint32_t>(nb, (want + ntile - 1) / ntile));
  const dim3 grid(ntile, gy);
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
                                    if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<L, NWV, WPSV, true>, pl.k1_fast_lds); \
                                    if (!rc && pl.k1_fast_lds + (size_t)(L >= 5 ? pl.r / 8 : pl.r) <= h->lds_limit) \
                                      rc = allow_lds(h, fast::frbch_k1_wave<L, NWV, WPSV, true, false, true>, pl.k1_fast_lds + (size_t)(L >= 5 ? pl.r / 8 : pl.r)); } while (0)
    rc = FRBCH_OK;
    if (pl.fast_k1_wave) switch (pl.fast_k1_log2m) {
      case 1: FRBCH_AL(1, 8, 1); break;
      case 2: FRBCH_AL(2, 8, 1); break;
      case 4:
        FRBCH_AL(4, 8, 2);
        if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<4, 8, 2, false, true>, pl.k1_fast_lds);
        if (!rc) rc = allow_lds(h, fast::frbch_k1_wave<4, 8, 2, true, true>, pl.k1_fast_lds);
        if (!rc && pl.k1_fast_lds + (size_t)pl.r <= h->lds_limit) rc = allow_lds(h, fast::frbch_k1_wave<4, 8, 2, true, true, true>, pl.k1_fast_lds + (size_t)pl.r);
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
// dynamic level setting: the low-state counts of the windows covering the launch's first `nsamples` samples (a multiple of the window)
int launch_dls_count(frbch_handle* h, KParams& p, uint64_t nsamples, dev_stream_t s) {
  const Plan& pl = h->pl;
  const uint64_t nwin = nsamples >> pl.dls_lg_ns;
  if (!h->dls_tab || (nwin << pl.dls_lg_ns) != nsamples) return fail(h, FRBCH_E_ARG, "dynamic level setting: the sample count is not a multiple of the window");
  if (nwin > h->dls_cap) {
    dev_free(h->dls_nlow);
    h->dls_nlow = nullptr;
    h->dls_cap = 0;
    CHECK_DEV(h, dev_malloc((void**)&h->dls_nlow, nwin * sizeof(uint32_t)), "hipMalloc(window counts)");
    h->dls_cap = nwin;
  }
  p.dls_tab = h->dls_tab;
  p.dls_nlow = h->dls_nlow;
  p.dls_lg_ns = pl.dls_lg_ns;
  KParams q = p;
  q.row0 = nwin;
  DEV_LAUNCH(frbch_dls_count, (nwin + 3) / 4, 1, 256, 2048, s, q);
  CHECK_DEV(h, dev_check_launch(), "launch window counts");
  return FRBCH_OK;
}

int launch_front(frbch_handle* h, KParams& p, uint32_t nb, dev_stream_t s, dev_stream_t sk) {
  const Plan& pl = h->pl;
  p.tile_major = 0;   // set by the K1 that writes that layout
  if (pl.dls_lg_ns) {
    const int rc = launch_dls_count(h, p, ((uint64_t)nb - 1) * pl.hop + pl.n, s);
    if (rc) return rc;
  }
  {
    // blocks that touch invalid / filler frames: the staged wave K1 reads their samples as 0 through a flag per row (MSK);
    // where it cannot run (R = 8192, unaligned input, the barrier K1) the generic K1 tests the bitmap per sample
    const bool masked = p.fbad != nullptr;
    if (!masked || pl.fast_k1_wave) {
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
    if (masked && !(pl.fast_k1_wave && (!pl.coherent || h->coh_order_m) && (done = launch_k1_fast(h, p, nb, s)))) {
      h->stg_ready = false;
      if (pl.coherent && h->coh_order_m) {   // from here on the generic K1 / K3 and their bin order
        const int rc = build_chirp(h, 0);
        if (rc) return rc;
      }
    } else if (masked) {
      // (the masked wave K1 ran)
    } else if (!pl.coherent) done = launch_k1_fast(h, p, nb, s);
    else if (h->coh_order_m) {
      done = launch_k1_fast(h, p, nb, s);
      if (!done) {   // a start offset the register kernel cannot gather: from here on the generic K1 / K3 and their bin order
        const int rc = build_chirp(h, 0);
        if (rc) return rc;
      }
    }
    if (!done) {
      DEV_LAUNCH(frbch_k1_branch, pl.c2 / pl.g, nb, pl.nthreads, pl.k1_lds, s, p);
      // the timing report says so when a launch of a handle planned for a register-pass K1 fell back to the generic one
      if (!h->kname[KID_K1].empty() && h->kname[KID_K1].find(kKernelNames[KID_K1]) == std::string::npos)
        h->kname[KID_K1] += std::string("+") + kKernelNames[KID_K1];
    }
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
      ProfScope ps(h, s, KID_K2, ((double)nb * 16.0 + (pl.coh_fast_c ? 1.0 : (double)nb) * 8.0) * (double)pl.n);   // the register K2c reads the kernel table once per launch
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
#ifndef FRBCH_NO_FAST
  const int kid = priv_takes(pl, p, h->priv_grid) ? KID_K2P : KID_K2;   // the two K2 families of 2C = 2048 report separately
#else
  const int kid = KID_K2;
#endif
  ProfScope ps(h, s, kid, bytes);
  if (!launch_k2_fast(h, p, nb, s)) DEV_LAUNCH(frbch_k2_chan, pl.r / tile_t, nb, pl.nthreads, pl.k2_lds, s, p);
  CHECK_DEV(h, dev_check_launch(), "launch K2");
  return FRBCH_OK;
}

// the table of partial rescale sums (separate statistics pass: partial_chunks rows; sums fused into K2: fused_chunks rows)
int ensure_partial(frbch_handle* h) {
  if (h->partial) return FRBCH_OK;
  const Plan& pl = h->pl;
  // rows of partial sums of the separate statistics pass = its workgroups (x threads sharing a column group).  Narrow rows (32 .. 128
  // columns: the online chain's channel counts) need more of them: 2048 rows left 256 single-wave workgroups for the whole chip and the
  // reduction took 11 % of a 32-channel step (0.45 ms for 40 MB); the table stays within 8 MB
  h->partial_chunks = (int)std::min<uint64_t>(32768, std::max<uint64_t>(2048, (8ull << 20) / (pl.ncol * 16)));
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
  if (!pow2 || pl.c < 4 || cg < 8) return false;   // (narrow rows, 32 .. 128 columns: a wave covers several rows -- one contiguous run all the same)
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

int run_quantise(frbch_handle* h, uint64_t rows, uint8_t* dst, dev_stream_t s, int ncu) {
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


int allow_generic_lds(frbch_handle* h) {
  const Plan& pl = h->pl;
  CHECK_DEV(h, dev_allow_lds(frbch_k1_branch, pl.k1_lds), "LDS size K1");
  CHECK_DEV(h, dev_allow_lds(frbch_k2_chan, pl.k2_lds), "LDS size K2");
  CHECK_DEV(h, dev_allow_lds(frbch_kc_dcfix, pl.kc_lds), "LDS size Kc");
  if (pl.coherent) {
    CHECK_DEV(h, dev_allow_lds(frbch_k2c_chirp, pl.k2_lds), "LDS size K2c");
    CHECK_DEV(h, dev_allow_lds(frbch_k3_dedisp, pl.k3_lds), "LDS size K3");
  }
  return FRBCH_OK;
}

int fused_chunks_of(const frbch_handle* h) {
#ifndef FRBCH_NO_FAST
  return fused_stat_chunks(h->pl, h->cfg.flags, h->cfg.pol_mode, h->priv_grid);
#else
  (void)h;
  return 0;
#endif
}

// the unpack tap: voltages as the filterbank sees them (A4 in isolation)
int launch_unpack_tap(frbch_handle* h, KParams& p, uint64_t nsamples, int decoder, dev_stream_t s) {
  if (h->pl.dls_lg_ns) {
    if (decoder != 0) return fail(h, FRBCH_E_ARG, "dynamic level setting is decoded by the generic unpack only");
    const int rc = launch_dls_count(h, p, nsamples, s);
    if (rc) return rc;
  }
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
  return FRBCH_OK;
}

}  // namespace frbchi
