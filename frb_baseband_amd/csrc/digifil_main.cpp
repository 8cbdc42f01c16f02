// `digifil` command-line shim: accepts the argv that run_digifil builds (process_vdif.py:156-182)
// and runs the MI355X channeliser instead of DSPSR.  Installed under the name `digifil` so that
// (a) the unmodified reference process_vdif.py drives it (process_vdif.py:191) and (b) the job
// throttle that counts `ps -ef | grep digifil` (base2fil.sh:21-28, online-deamon.sh:50) still sees
// one process per IF.  Exit status 0 on success; non-zero with the reason on stderr otherwise,
// which the caller turns into RunError (process_vdif.py:193-198).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/frbch.h"

int main(int argc, char** argv) {
  frbch_config cfg;
  frbch_config_init(&cfg);
  char hdr[1024] = "", out[1024] = "", err[512] = "";
  int rc = frbch_parse_digifil_argv(argc, (const char* const*)argv, &cfg, hdr, sizeof hdr, out, sizeof out, err,
                                    sizeof err);
  if (rc) {
    fprintf(stderr, "digifil(frbch): %s: %s\n", frbch_strerror(rc), err);
    return 1;
  }
  rc = frbch_config_from_hdr(hdr, &cfg);
  if (rc) {
    fprintf(stderr, "digifil(frbch): cannot use header %s: %s\n", hdr, frbch_strerror(rc));
    return 1;
  }
  const char* dev = getenv("FRBCH_DEVICE");  // GPU ordinal for this IF (set by the multi-IF launcher)
  if (dev) cfg.device = atoi(dev);
  frbch_handle* h = NULL;
  rc = frbch_open(&cfg, &h);
  if (rc) {
    fprintf(stderr, "digifil(frbch): %s: %s\n", frbch_strerror(rc), h ? frbch_last_error(h) : "");
    frbch_close(h);
    return 1;
  }
  rc = frbch_run_file(h, cfg.datafile, out);
  if (rc) {
    fprintf(stderr, "digifil(frbch): %s: %s\n", frbch_strerror(rc), frbch_last_error(h));
    frbch_close(h);
    return 1;
  }
  frbch_info info;
  frbch_get_info(h, &info);
  fprintf(stdout, "digifil(frbch): wrote %llu samples x %u ch x %u if to %s (%llu blocks)\n",
          (unsigned long long)info.rows_out, info.nchan, info.nif, out, (unsigned long long)info.blocks_done);
  frbch_close(h);
  return 0;
}
