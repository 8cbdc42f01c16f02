// Host sinks of the product path, measured alone (no GPU): how fast can ONE sequential writer push bytes into what the
// reference's pipeline reads from?  The reference lets digifil write into a named FIFO that sigproc `splice` drains
// (base2fil.sh:348-350, 417-446), the FIFOs enlarged to 1 MiB (setfifo.perl:10); frbch_run_file / frbch_run_scan write the same
// way (one strictly sequential writer) or into a regular file.  Ceilings printed in GB/s for a buffer of the size of a pinned slot:
//   fifo_write     write() into a 1-MiB FIFO drained by a reader thread that read()s into a 16-MiB buffer (what splice does)
//   fifo_vmsplice  vmsplice() of the same user pages into the FIFO (no copy on the writer's side; the pages must stay untouched
//                  until the reader has taken them: only usable with a ring at least as deep as the pipe plus one slot)
//   fifo_64k       the same write() with the default 64-KiB pipe (no F_SETPIPE_SZ)
//   tmpfs_write    write() into a fresh file on /dev/shm
//   tmpfs_prealloc the same after posix_fallocate (what run_pipelined does before its parallel copies)
//   devnull        write() into /dev/null (the syscall cost alone)
// g++ -O2 -pthread tools/micro/pipe_bench.cpp -o tools/micro/pipe_bench && tools/micro/pipe_bench [MiB total, default 2048]
#define _GNU_SOURCE 1
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

static double now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static bool write_all(int fd, const uint8_t* p, size_t n) {
  while (n) {
    const ssize_t w = write(fd, p, n);
    if (w < 0) { if (errno == EINTR) continue; return false; }
    p += w; n -= (size_t)w;
  }
  return true;
}
static bool vmsplice_all(int fd, uint8_t* p, size_t n) {
  while (n) {
    struct iovec iov = {p, n};
    const ssize_t w = vmsplice(fd, &iov, 1, 0);
    if (w < 0) { if (errno == EINTR) continue; return false; }
    p += w; n -= (size_t)w;
  }
  return true;
}

static double fifo_run(const std::string& path, size_t total, size_t slot, int pipe_sz, bool use_vmsplice, int nslots) {
  unlink(path.c_str());
  if (mkfifo(path.c_str(), 0600) != 0) return -1;
  std::thread reader([&]() {
    const int fd = open(path.c_str(), O_RDONLY);
    std::vector<uint8_t> buf(16u << 20);
    while (read(fd, buf.data(), buf.size()) > 0) {}
    close(fd);
  });
  const int fd = open(path.c_str(), O_WRONLY);
  if (pipe_sz) (void)fcntl(fd, F_SETPIPE_SZ, pipe_sz);
  std::vector<std::vector<uint8_t>> ring(nslots, std::vector<uint8_t>(slot, 1));
  const double t0 = now();
  size_t done = 0, i = 0;
  bool ok = true;
  while (done < total && ok) {
    uint8_t* p = ring[i % nslots].data();
    ok = use_vmsplice ? vmsplice_all(fd, p, slot) : write_all(fd, p, slot);
    done += slot;
    ++i;
  }
  close(fd);
  reader.join();
  const double dt = now() - t0;
  unlink(path.c_str());
  return ok ? done / dt / 1e9 : -1;
}

static double file_run(const std::string& path, size_t total, size_t slot, bool prealloc) {
  unlink(path.c_str());
  const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  if (fd < 0) return -1;
  std::vector<uint8_t> buf(slot, 1);
  const double t0 = now();
  if (prealloc && posix_fallocate(fd, 0, (off_t)total) != 0) { close(fd); return -1; }
  size_t done = 0;
  bool ok = true;
  while (done < total && ok) { ok = write_all(fd, buf.data(), slot); done += slot; }
  close(fd);
  const double dt = now() - t0;
  unlink(path.c_str());
  return ok ? done / dt / 1e9 : -1;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20;
  const size_t slot = 16u << 20;
  const std::string base = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : "/tmp";
  const std::string fifo = base + "/frbch_pipe_bench.fifo", file = base + "/frbch_pipe_bench.bin";
  printf("{\"bytes\": %zu, \"slot\": %zu", total, slot);
  for (int rep = 0; rep < 2; ++rep) {   // (second pass: warm)
    const double a = fifo_run(fifo, total, slot, 1 << 20, false, 2);
    const double b = fifo_run(fifo, total, slot, 1 << 20, true, 8);
    const double c = fifo_run(fifo, total, slot, 0, false, 2);
    const double d = file_run(file, std::min<size_t>(total, (size_t)1 << 30), slot, false);
    const double e = file_run(file, std::min<size_t>(total, (size_t)1 << 30), slot, true);
    const int nfd = open("/dev/null", O_WRONLY);
    std::vector<uint8_t> buf(slot, 1);
    const double t0 = now();
    for (size_t done = 0; done < total; done += slot) write_all(nfd, buf.data(), slot);
    const double f = total / (now() - t0) / 1e9;
    close(nfd);
    if (rep == 1)
      printf(", \"fifo_write_GBs\": %.2f, \"fifo_vmsplice_GBs\": %.2f, \"fifo_64k_GBs\": %.2f, \"tmpfs_write_GBs\": %.2f, \"tmpfs_prealloc_GBs\": %.2f, \"devnull_GBs\": %.1f",
             a, b, c, d, e, f);
  }
  printf("}\n");
  return 0;
}
