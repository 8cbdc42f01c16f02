// libfrbch: the C ABI of include/frbch.h that works on device memory -- life cycle, rescale state, device entry points, timing
// (frbch_internal.h lists the units).
#include "frbch_internal.h"

// =============================================================================================
// C ABI
// =============================================================================================
// Work the device entry points queued on a caller's stream (statistics writing offset / scale, the digitiser reading
// them, the power buffer) must be complete before the handle's own stream or the host touches that state.
// The caller's stream may be gone by then (a temporary stream of the caller's framework): what is waited for is an event of
// the handle's own, recorded on that stream behind the call's work (mark_user_stream).
namespace frbchi {
int settle_user_stream(frbch_handle* h) {
  if (h->user_stream) {
    h->user_stream = 0;                     // (cleared first: a failing wait must not wedge every later call)
    if (h->user_ev_made && dev_event_sync(h->user_ev) != 0)
      return fail(h, FRBCH_E_DEVICE, std::string("sync (caller stream): ") + dev_last_error_string());
  }
  return FRBCH_OK;
}
void mark_user_stream(frbch_handle* h, dev_stream_t s) {
  if (!h->user_ev_made) {
    if (dev_event_create_sync(&h->user_ev) != 0) return;
    h->user_ev_made = true;
  }
  dev_event_record(h->user_ev, s);
  h->user_stream = s;
}
}  // namespace frbchi
using namespace frbchi;

#ifdef FRBCH_EXPERIMENTS
extern "C" const char* frbch_version(void) { return "frbch abi 5 backend " FRBCH_BACKEND_NAME " +experiments"; }
#else
extern "C" const char* frbch_version(void) { return "frbch abi 5 backend " FRBCH_BACKEND_NAME; }
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
  if ((cfg->flags & kFlagBuffered) && (cfg->flags & kFlagTwoPass)) return fail(h, FRBCH_E_ARG, "cfg.flags asks for the buffered AND the two-pass rescale");
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
  if (h->cfg.unpack_mode == 1) h->cfg.flags |= kFlagGenericK1;   // the per-window levels are looked up by the generic K1 only
  const std::string why = make_plan(h->cfg, &h->pl, h->lds_limit);
  if (!why.empty()) return fail(h, FRBCH_E_ARG, why);
  const Plan& pl = h->pl;
  CHECK_DEV(h, dev_stream_create(&h->stream), "hipStreamCreate");
  { const int rc0 = allow_generic_lds(h); if (rc0) return rc0; }

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
    if (pl.fast_k2_priv) {   // (KID_K2 keeps frbch_k2_wave's name: float rows of four products and fallen-back launches run it)
      const int pmn = h->cfg.pol_mode == 2 ? 2 : (h->cfg.pol_mode >= 4 ? h->cfg.pol_mode : 0);
      snprintf(nm, sizeof nm, "frbch_k2_priv<%d>", pmn);
      h->kname[KID_K2P] = nm;
      snprintf(nm, sizeof nm, "frbch_k2_priv<%d,stats>", pmn);
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
    CHECK_DEV(h, dev_malloc((void**)&h->spill2, (size_t)pl.maxb * pl.n * sizeof(cf)), "hipMalloc(spill2)");
    CHECK_DEV(h, dev_malloc((void**)&h->chirp, (size_t)pl.n * sizeof(cf)), "hipMalloc(chirp)");
    CHECK_DEV(h, dev_malloc((void**)&h->ptmp, (size_t)pl.maxb * pl.rows_per_block * pl.ncol * sizeof(float)), "hipMalloc(ptmp)");
    if ((rc = build_chirp(h, pl.coh_fast_r ? (1 << pl.coh_fast_r) : 0))) return rc;
    h->kname[KID_K2] = pl.coh_fast_c ? "frbch_k2c_fast" : "frbch_k2c_chirp";
    if (pl.ncol % 64 == 0 && pl.rows_per_block % 2 == 0 && pl.c % 4 == 0 && !(h->cfg.flags & 2u)) h->kname[KID_K4] = "frbch_k4_fast";
    if (pl.coh_fast_r) h->kname[KID_K3] = (pl.coh_fast_r == 4 && pl.coh_nt == 512 && !(h->cfg.flags & 8u)) ? "frbch_k3_wave<4>" : "frbch_k3_fast";
  }
  if (pl.dls_lg_ns) {
    const std::vector<float> tab = dls_table(1u << pl.dls_lg_ns, h->cfg.dls_cutoff_sigma, h->cfg.dls_threshold);
    CHECK_DEV(h, dev_malloc((void**)&h->dls_tab, tab.size() * sizeof(float)), "hipMalloc(level table)");
    CHECK_DEV(h, dev_h2d(h->dls_tab, tab.data(), tab.size() * sizeof(float), h->stream), "upload level table");
    CHECK_DEV(h, dev_sync(h->stream), "sync");
    h->dls_cap = (((uint64_t)pl.maxb - 1) * pl.hop + pl.n) >> pl.dls_lg_ns;
    CHECK_DEV(h, dev_malloc((void**)&h->dls_nlow, h->dls_cap * sizeof(uint32_t)), "hipMalloc(window counts)");
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
  dev_free(h->dls_tab); dev_free(h->dls_nlow);
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
  { const int rc = launch_unpack_tap(h, p, nsamples, decoder, s); if (rc) return rc; }
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

