// libfrbch: stream state of a handle -- batches, the rescale-interval state machine (buffered and two-pass forms), the chain of
// stages of a scan and the two kernels that may share the chip (frbch_internal.h lists the units).
#include "frbch_internal.h"

namespace frbchi {

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
// (the flush, or the next call, then finds the batch deferred).  Automatic for four products (measured, config 3: 35.3 ms per 8-IF
// step against 36.0 buffered); one product stays buffered (its float rows are a quarter of the bytes: 3.3
// against 3.9 ms at config 2).  flags: 1 << 27 forces the buffered form, 1 << 28 the two-pass form.
bool twopass_usable(const frbch_handle* h) {
  const Plan& pl = h->pl;
  if (!(pl.fast_k2_priv && h->priv_grid > 0 && h->cfg.rescale_constant && pl.interval_rows > 0 && !pl.k2_two_stage && fused_chunks_of(h) > 0))
    return false;
  if (h->cfg.flags & kFlagBuffered) return false;
  return (h->cfg.flags & kFlagTwoPass) || pl.nif == 4;
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

std::mutex g_lanes_mutex;
std::vector<std::pair<std::pair<int, int>, Lanes*>> g_lanes;   // (device, front CUs) -> lanes; live until the process ends

Lanes* get_lanes(int device, int ncu_front, bool plain) {
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
                dev_stream_t s, const uint8_t* h_bad, uint64_t nfr_bad, const uint32_t* d_fbad, Chain* chain) {
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

int engine_flush(frbch_handle* h, uint8_t* d_out, size_t cap, uint64_t* rows_written, dev_stream_t s, Chain* ch) {
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


}  // namespace frbchi
