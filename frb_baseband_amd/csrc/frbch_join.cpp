// frbch_join -- streaming frequency concatenation of SIGPROC filterbanks: what sigproc `splice` does at the end of a scan
// (base2fil.sh:404-448: `splice <IF_n.fil> ... <IF_1.fil> > IFall.fil`, inputs listed highest frequency first, :350,367), as
// a native program that is fed WHILE the producers run.  The node-level scan (frb_baseband_amd/scan.py) starts one process
// per GPU, each writing its IFs -- already joined on its GPU by frbch_run_scan -- into a named FIFO; this program reads
// the FIFOs in lock step and writes ONE file, strictly sequentially (the output may itself be a FIFO: no seek, no O_EXCL,
// never unlinked, INSTALL.md:32-35).
//
//   frbch_join <out.fil> <piece_0> [<piece_1> ...]          pieces in descending frequency, any mix of files and FIFOs
//
// Output: the SIGPROC header of piece 0 with nchans = sum over the pieces, then rows [t][product][piece-major channels];
// rows are cut to the shortest piece, as splice does.  nbits / nifs / tsamp must agree, tstart within half a sample, and the
// pieces must continue each other in frequency in the order given (same foff, fch1 of piece i = fch1 + nchans foff of piece
// i - 1): a wrong order would otherwise produce a silently mislabelled IFall file.
// One reader thread per piece (two row blocks in flight each) so that every producer is drained while the join writes.
// Exit status 0 on success, 1 with the reason on stderr otherwise; a failing piece fails the run.
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

bool read_exact(int fd, uint8_t* p, size_t n, size_t* got) {   // false on error; *got < n on end of file
  *got = 0;
  while (*got < n) {
    const ssize_t r = read(fd, p + *got, n - *got);
    if (r < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    if (r == 0) break;
    *got += (size_t)r;
  }
  return true;
}
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

struct Header {
  std::vector<uint8_t> bytes;   // HEADER_START .. HEADER_END as read
  size_t nchans_value_off = 0;  // offset of the 4-byte nchans value inside bytes
  int nchans = 0, nbits = 0, nifs = 1;
  double tsamp = 0.0, tstart = 0.0;
  double fch1 = 0.0, foff = 0.0;   // centre of channel 0, channel step (MHz)
  bool have_freq = false;
};

// keyword types of the SIGPROC header (the set frbch_host.cpp writes, plus the other standard ones)
int key_kind(const std::string& k) {   // 1 int, 2 double, 3 string, 0 none (markers), -1 unknown
  static const char* ints[] = {"telescope_id", "machine_id", "data_type", "barycentric", "pulsarcentric", "nbits", "nsamples",
                               "nchans", "nifs", "nbeams", "ibeam"};
  static const char* dbls[] = {"az_start", "za_start", "src_raj", "src_dej", "tstart", "tsamp", "fch1", "foff", "refdm", "period"};
  static const char* strs[] = {"rawdatafile", "source_name"};
  if (k == "HEADER_START" || k == "HEADER_END") return 0;
  for (const char* s : ints) if (k == s) return 1;
  for (const char* s : dbls) if (k == s) return 2;
  for (const char* s : strs) if (k == s) return 3;
  return -1;
}

// reads exactly the header from a stream (no read-ahead into the samples: the input may be a FIFO)
bool read_header(int fd, Header* h, std::string* why) {
  auto need = [&](size_t n) -> bool {
    const size_t old = h->bytes.size();
    h->bytes.resize(old + n);
    size_t got = 0;
    if (!read_exact(fd, h->bytes.data() + old, n, &got) || got != n) {
      *why = "short read inside the SIGPROC header";
      return false;
    }
    return true;
  };
  auto rd_str = [&](std::string* s) -> bool {
    if (!need(4)) return false;
    int32_t n;
    memcpy(&n, h->bytes.data() + h->bytes.size() - 4, 4);
    if (n < 0 || n > 255) { *why = "bad string length in the SIGPROC header"; return false; }   // (values may be empty)
    if (n && !need((size_t)n)) return false;
    s->assign((const char*)h->bytes.data() + h->bytes.size() - (size_t)n, (size_t)n);
    return true;
  };
  std::string key;
  if (!rd_str(&key) || key != "HEADER_START") { if (why->empty()) *why = "not a SIGPROC filterbank (no HEADER_START)"; return false; }
  for (;;) {
    if (!rd_str(&key)) return false;
    if (key == "HEADER_END") break;
    const int kind = key_kind(key);
    if (kind == 1) {
      if (!need(4)) return false;
      int32_t v;
      memcpy(&v, h->bytes.data() + h->bytes.size() - 4, 4);
      if (key == "nchans") { h->nchans = v; h->nchans_value_off = h->bytes.size() - 4; }
      else if (key == "nbits") h->nbits = v;
      else if (key == "nifs") h->nifs = v;
    } else if (kind == 2) {
      if (!need(8)) return false;
      double v;
      memcpy(&v, h->bytes.data() + h->bytes.size() - 8, 8);
      if (key == "tsamp") h->tsamp = v;
      else if (key == "tstart") h->tstart = v;
      else if (key == "fch1") { h->fch1 = v; h->have_freq = true; }
      else if (key == "foff") h->foff = v;
    } else if (kind == 3) {
      std::string v;
      if (!rd_str(&v)) return false;
    } else {
      *why = "unknown SIGPROC header key '" + key + "'";
      return false;
    }
  }
  if (h->nchans <= 0 || h->nbits <= 0 || h->nifs <= 0 || !h->nchans_value_off) { *why = "SIGPROC header without nchans / nbits"; return false; }
  if ((h->nchans * h->nbits) % 8) { *why = "a product line of a piece is not a whole number of bytes"; return false; }
  return true;
}

struct Piece {
  int fd = -1;
  Header hdr;
  size_t seg = 0;                 // bytes of one (row, product) line of this piece
  size_t row_bytes = 0;           // seg * nifs
  // two row blocks in flight
  std::vector<uint8_t> buf[2];
  size_t rows[2] = {0, 0};        // whole rows in buf[k]
  int state[2] = {0, 0};          // 0 free, 1 filled
  bool eof = false, failed = false;
  int err = 0;                    // errno of the failed read (errno is thread-local: the reader thread keeps it here)
  std::mutex m;
  std::condition_variable cv;
  std::thread th;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: frbch_join <out.fil> <piece_0> [<piece_1> ...]   (pieces in descending frequency)\n");
    return 1;
  }
  const int np = argc - 2;
  std::vector<Piece> pc(np);
  for (int i = 0; i < np; ++i) {
    pc[i].fd = open(argv[2 + i], O_RDONLY);   // (a FIFO: blocks until its producer opens it)
    if (pc[i].fd < 0) { fprintf(stderr, "frbch_join: cannot open %s: %s\n", argv[2 + i], strerror(errno)); return 1; }
#ifdef F_SETPIPE_SZ
    (void)fcntl(pc[i].fd, F_SETPIPE_SZ, 1 << 20);   // FIFOs: 1 MiB in flight (best effort; fails harmlessly on files)
#endif
  }
  size_t line_out = 0;
  for (int i = 0; i < np; ++i) {
    std::string why;
    if (!read_header(pc[i].fd, &pc[i].hdr, &why)) { fprintf(stderr, "frbch_join: %s: %s\n", argv[2 + i], why.c_str()); return 1; }
    const Header &a = pc[i].hdr, &b = pc[0].hdr;
    if (a.nbits != b.nbits || a.nifs != b.nifs || a.tsamp != b.tsamp) { fprintf(stderr, "frbch_join: %s: nbits / nifs / tsamp differ from the first piece\n", argv[2 + i]); return 1; }
    if (fabs(a.tstart - b.tstart) > 0.5 * b.tsamp / 86400.0) { fprintf(stderr, "frbch_join: %s: tstart differs from the first piece\n", argv[2 + i]); return 1; }
    // the output keeps piece 0's header (fch1, foff) with nchans patched: that labels the channels correctly only when the pieces
    // continue each other in frequency, in the order given: fch1_i = fch1_{i-1} + nchans_{i-1} foff, same foff (sign included)
    if (i > 0 && a.have_freq && pc[i - 1].hdr.have_freq) {
      const Header& q = pc[i - 1].hdr;
      const double tol = 1e-6 * fabs(q.foff) + 1e-9;
      if (fabs(a.foff - q.foff) > tol) { fprintf(stderr, "frbch_join: %s: foff %.9g differs from the previous piece's %.9g\n", argv[2 + i], a.foff, q.foff); return 1; }
      const double want = q.fch1 + (double)q.nchans * q.foff;
      if (fabs(a.fch1 - want) > 0.01 * fabs(q.foff)) {
        fprintf(stderr, "frbch_join: %s: fch1 %.9g MHz does not continue the previous piece (expected %.9g): pieces out of order or not contiguous\n", argv[2 + i], a.fch1, want);
        return 1;
      }
    }
    pc[i].seg = (size_t)a.nchans * (size_t)a.nbits / 8;
    pc[i].row_bytes = pc[i].seg * (size_t)a.nifs;
    line_out += pc[i].seg;
  }
  const int nifs = pc[0].hdr.nifs;
  const size_t row_out = line_out * (size_t)nifs;
  // rows per block: ~8 MiB of output
  const size_t block_rows = std::max<size_t>(1, ((size_t)8 << 20) / row_out);

  const int out_fd = open(argv[1], O_WRONLY | O_CREAT | O_TRUNC, 0644);   // no O_EXCL: may be a pre-made FIFO
  if (out_fd < 0) { fprintf(stderr, "frbch_join: cannot open %s: %s\n", argv[1], strerror(errno)); return 1; }
  {
    std::vector<uint8_t> hdr = pc[0].hdr.bytes;
    int32_t total = 0;
    for (int i = 0; i < np; ++i) total += pc[i].hdr.nchans;
    memcpy(hdr.data() + pc[0].hdr.nchans_value_off, &total, 4);
    if (!write_all(out_fd, hdr.data(), hdr.size())) { fprintf(stderr, "frbch_join: write: %s\n", strerror(errno)); return 1; }
  }

  for (int i = 0; i < np; ++i) {
    Piece* p = &pc[i];
    p->buf[0].resize(block_rows * p->row_bytes);
    p->buf[1].resize(block_rows * p->row_bytes);
    p->th = std::thread([p, block_rows]() {
      for (int k = 0;; k ^= 1) {
        {
          std::unique_lock<std::mutex> lk(p->m);
          p->cv.wait(lk, [&] { return p->state[k] == 0; });
        }
        size_t got = 0;
        const bool ok = read_exact(p->fd, p->buf[k].data(), block_rows * p->row_bytes, &got);
        const int err = ok ? 0 : errno;
        std::lock_guard<std::mutex> lk(p->m);
        if (!ok) p->err = err;
        p->rows[k] = got / p->row_bytes;            // (a trailing partial row is dropped, as splice does)
        p->state[k] = 1;
        if (!ok) p->failed = true;
        if (!ok || got < block_rows * p->row_bytes) p->eof = true;
        p->cv.notify_all();
        if (p->eof) return;
      }
    });
  }

  std::vector<uint8_t> out(block_rows * row_out);
  uint64_t rows_written = 0;
  int rc = 0;
  bool done = false;
  for (int k = 0; !done; k ^= 1) {
    size_t rows = block_rows;
    for (int i = 0; i < np; ++i) {
      Piece& p = pc[i];
      std::unique_lock<std::mutex> lk(p.m);
      p.cv.wait(lk, [&] { return p.state[k] == 1; });
      if (p.failed) { fprintf(stderr, "frbch_join: read %s: %s\n", argv[2 + i], strerror(p.err)); rc = 1; }
      rows = std::min(rows, p.rows[k]);
      if (p.rows[k] < block_rows) done = true;      // the shortest piece ends the output
    }
    if (rc) break;
    size_t off = 0;
    for (int i = 0; i < np; ++i) {
      const Piece& p = pc[i];
      for (size_t r = 0; r < rows * (size_t)nifs; ++r) memcpy(out.data() + r * line_out + off, p.buf[k].data() + r * p.seg, p.seg);
      off += p.seg;
    }
    if (rows && !write_all(out_fd, out.data(), rows * row_out)) { fprintf(stderr, "frbch_join: write: %s\n", strerror(errno)); rc = 1; break; }
    rows_written += rows;
    for (int i = 0; i < np; ++i) {
      Piece& p = pc[i];
      std::lock_guard<std::mutex> lk(p.m);
      p.state[k] = 0;
      p.cv.notify_all();
    }
  }
  // let every producer finish: drain what longer pieces still send (their writers would block on a full pipe otherwise)
  for (int i = 0; i < np; ++i) {
    Piece& p = pc[i];
    for (;;) {
      std::unique_lock<std::mutex> lk(p.m);
      if (p.eof) {
        p.state[0] = p.state[1] = 0;
        p.cv.notify_all();
        break;
      }
      p.state[0] = p.state[1] = 0;
      p.cv.notify_all();
      p.cv.wait(lk, [&] { return p.state[0] == 1 || p.state[1] == 1 || p.eof; });
    }
    p.th.join();
    close(p.fd);
  }
  if (close(out_fd) != 0 && !rc) { fprintf(stderr, "frbch_join: close: %s\n", strerror(errno)); rc = 1; }
  if (!rc) fprintf(stdout, "frbch_join: wrote %llu rows x %d products x %zu bytes from %d pieces to %s\n", (unsigned long long)rows_written, nifs, line_out, np, argv[1]);
  return rc;
}
