// libfrbch: the C ABI of include/frbch.h that works on host memory and files -- push / flush / pull, the pipelined whole-file path
// (what `digifil ... -o <out> <hdr>` does, process_vdif.py:191) and the scan of several IFs into one file (base2fil.sh:348-350,
// 404-448) (frbch_internal.h lists the units).
#include "frbch_internal.h"

using namespace frbchi;

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
    { const int rc0 = allow_generic_lds(h); if (rc0) return rc0; }
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

// A FIFO sink (base2fil.sh:348-350: digifil writes into a named pipe that sigproc `splice` read()s, enlarged to 1 MiB by
// setfifo.perl:10).  write() copies every byte into pipe pages; vmsplice() hands the pages of the (pinned) slot to the pipe BY
// REFERENCE -- measured alone, tools/micro/pipe_bench: 8.6 against 5.9 GB/s for a 16-MiB slot into a drained 1-MiB FIFO.  Pages given
// by reference must stay untouched until the reader has taken them, so the LAST pipe's worth of every buffer is copied with write():
// the pipe is a ring of pipe_size / page buffers, and when that write has returned the ring holds nothing but its copies -- every
// page handed over before it has been consumed and the caller may refill the buffer at once (no deeper ring, no bookkeeping).
// Falls back to write() for good when the first vmsplice is refused (not a pipe, pages that cannot be referenced), and with
// FRBCH_FIFO_COPY=1 in the environment (for a reader that forwards the pages by reference itself instead of read()ing them).
struct FifoSink {
  size_t pipe_bytes = 0;     // 0 = not a FIFO
  bool by_reference = false;
  uint64_t referenced = 0;   // bytes that went out by reference (frbch_info::diag bit 1 says whether any did)
};
FifoSink probe_sink(int fd) {
  FifoSink f;
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0 || !S_ISFIFO(st.st_mode)) return f;
#ifdef F_GETPIPE_SZ
  const int sz = fcntl(fd, F_GETPIPE_SZ);
  if (sz > 0) f.pipe_bytes = (size_t)sz;
#endif
  const char* e = getenv("FRBCH_FIFO_COPY");
  f.by_reference = f.pipe_bytes >= 65536 && !(e && *e && *e != '0');
  return f;
}
bool sink_write(int fd, const uint8_t* p, size_t n, FifoSink* f) {
  if (f && f->by_reference && n > 2 * f->pipe_bytes) {
    const size_t nv = (n - f->pipe_bytes) & ~(size_t)4095;
    size_t left = nv;
    while (left) {
      struct iovec iov = {const_cast<uint8_t*>(p) + (nv - left), left};
      const ssize_t w = vmsplice(fd, &iov, 1, 0);
      if (w < 0) {
        if (errno == EINTR) continue;
        if (left == nv && (errno == EFAULT || errno == EINVAL || errno == EBADF || errno == ENOSYS || errno == EPERM || errno == ENOMEM)) {
          f->by_reference = false;     // nothing of this buffer has gone out yet: copy it, and every later one
          break;
        }
        return false;
      }
      left -= (size_t)w;
      f->referenced += (uint64_t)w;
    }
    if (f->by_reference) {
      p += nv;
      n -= nv;
    }
  }
  return write_all(fd, p, n);
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
  h0->diag &= ~3u;
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
  FifoSink fsink = probe_sink(out_fd);              // (one writer when the output is not a preallocated regular file)
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
          ok = sink_write(out_fd, outbuf[slot], n, &fsink);
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
  if (fsink.referenced) h0->diag |= 2u;   // frbch_info::diag bit 1: a FIFO took pages of the pinned ring by reference (vmsplice)
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
  FifoSink fsink = probe_sink(fd);
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
      if (!sink_write(fd, stage, nr * row_pitch, &fsink)) return fail(h0, FRBCH_E_IO, std::string("write: ") + strerror(errno));
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

