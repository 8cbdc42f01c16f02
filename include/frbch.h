/* frbch.h -- C ABI of the MI355X VDIF -> SIGPROC-filterbank channeliser.
 *
 * Drop-in boundary.  The reference (pharaofranz/frb-baseband) has no FFI for this path: the
 * boundary is a subprocess plus two files -- process_vdif.py:157-182 builds a `digifil` argv,
 * :191 launches it, the input is the `.hdr` written by make_hdr (:115-139) whose DATAFILE is the
 * per-IF VDIF, the output is the SIGPROC `.fil` named at :143-145 (possibly a FIFO,
 * base2fil.sh:348-350).  Each entry point below cites the piece of that interface it replaces.
 * Plain pointers and sizes only; no C++/torch types.  Handles are single-threaded, one per IF
 * (base2fil.sh:60-66 runs one process per IF).  The library never calls exit().
 *
 * There is NO CPU fallback: every compute entry point needs a gfx950 device and fails with
 * FRBCH_E_DEVICE otherwise.
 */
#ifndef FRBCH_H
#define FRBCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRBCH_ABI_VERSION 5

/* error codes (negative); 0 = ok.  process_vdif.py:193-198 turns a non-zero digifil exit status
 * into RunError; the CLI shim maps any of these to exit status 1 with frbch_strerror on stderr. */
enum {
  FRBCH_OK = 0,
  FRBCH_E_ARG = -1,      /* bad argument / unsupported configuration (InputError territory)   */
  FRBCH_E_IO = -2,       /* open/read/write failure                                           */
  FRBCH_E_FORMAT = -3,   /* not the VDIF / .hdr we understand                                 */
  FRBCH_E_DEVICE = -4,   /* no usable GPU, HIP error                                          */
  FRBCH_E_NOMEM = -5,
  FRBCH_E_STATE = -6,    /* call sequence error (e.g. pull before any push)                   */
  FRBCH_E_CAPACITY = -7  /* caller buffer too small                                           */
};

/* pol_mode: what run_digifil maps --pol to (process_vdif.py:163-176):
 *   0,1 -> -P0/-P1 single polarisation power; 2 -> -d1 PP+QQ; 3 -> -d3 (PP+QQ)^2;
 *   4 -> -d4 PP,QQ,Re(PQ*),Im(PQ*)  (help text :58-64; base2fil.sh:214-217);
 *   5 -> Stokes I,Q,U,V formed from the -d4 products for the circular feeds make_hdr declares ("BASIS Circular",
 *        process_vdif.py:131): I = PP+QQ, Q = 2 Re(PQ*), U = 2 Im(PQ*), V = PP-QQ.  An extension (north_star "IQUV
 *        formation"): the reference's --pol stops at 4 (:175-176); the shim spells it `-d4 -iquv`. */
typedef struct frbch_config {
  uint32_t size;               /* = sizeof(frbch_config); versioning                          */
  uint32_t abi_version;        /* = FRBCH_ABI_VERSION                                         */
  double freq_mhz;             /* .hdr FREQ (centre, process_vdif.py:127)                     */
  double bw_mhz;               /* .hdr BW, signed: < 0 = LSB (process_vdif.py:117-118,128)    */
  double start_s;              /* -S (process_vdif.py:157,160)                                */
  double total_s;              /* -T                                                          */
  uint32_t nchan;              /* -F<nchan>:...  (process_vdif.py:162-171)                    */
  uint32_t freq_res;           /* -F...:<freq_res>; 0 = 512 if nchan<=128 else 2*nchan (:162) */
  uint32_t tscrunch;           /* -t (only passed when > 1, process_vdif.py:156-158)          */
  int32_t nbit_out;            /* -b: 2, 8, 16, -32 (process_vdif.py:153-155)                 */
  int32_t pol_mode;            /* see above                                                   */
  uint32_t rescale_constant;   /* -c (always passed, process_vdif.py:157,160)                 */
  double rescale_interval_s;   /* -I secs; 0 = -I0 = keepBP (process_vdif.py:181-182)         */
  double dm;                   /* -D (last one wins; header refdm; process_vdif.py:177-178)   */
  uint32_t coherent;           /* -F<nchan>:D (process_vdif.py:179-180): dedisperse in the filterbank */
  int32_t device;              /* GPU ordinal (>= 0)                                          */
  uint32_t max_blocks_per_launch; /* 0 = auto; filterbank blocks batched per kernel launch    */
  uint32_t flags;              /* 0 in production.  Four kernel-selection switches, every one produces the same (correct) output and
                                * has parity cases: 1 generic (radix-2) K1, 2 generic K2 and back end -- the cross-check family of
                                * tests/test_gpu_stress.py --, 1<<20 rescale statistics in a separate pass over the buffered power
                                * rows instead of inside K2, and the form of a first `-c` rescale interval (bits 27 / 28: neither =
                                * automatic, 1<<27 buffered -- float rows written, then digitised --, 1<<28 two-pass -- K2 runs twice
                                * over the resident spill, no float rows; automatic = two-pass for four products at 1024 channels,
                                * DESIGN.md section 5b; both at once is an error).
                                * Any other bit makes frbch_open fail with FRBCH_E_ARG: rejected kernel variants, layouts and lane
                                * modes kept for A/B runs and the timing-only ablations (which produce WRONG output) exist only in
                                * libraries built with -DFRBCH_EXPERIMENTS (make EXPERIMENTS=1), never in the product build. */
  char telescope[64];          /* .hdr TELESCOPE  (process_vdif.py:123)                       */
  char source[64];             /* .hdr SOURCE     (:124)                                      */
  char ra[32];                 /* .hdr RA         (:125)                                      */
  char dec[32];                /* .hdr DEC        (:126)                                      */
  char datafile[512];          /* .hdr DATAFILE   (:129)                                      */
  uint32_t input_bits;         /* bits per sample of the VDIF: 2, or 1 (mode VDIF_8000-1024-16-1, spif2file.sh:58-61);
                                * 0 = take it from the first frame header (host paths) / 2 (device paths)            */
  uint32_t overlap;            /* 0 in production (automatic).  The digitiser of a completed rescale interval may run beside the K1 of
                                * the next IF of a scan, on plain streams, holding its CUs by an LDS reservation (DESIGN.md section
                                * 4b; automatic: four products at 8 bits, 11/16 of the CUs for K1; else off).  1 = off: every kernel
                                * on the whole chip, one after the other.  (3 << 24) | n = that mode with n CUs (a multiple of 8)
                                * left to K1.  Every setting produces the same output.  CU-masked lane modes (1 << 24, 2 << 24) and
                                * forced batching (bits 16..23) were measured slower and exist only in FRBCH_EXPERIMENTS builds. */
  float levels[4];             /* 2-bit level table, state 0..3 -> voltage (process_vdif.py:157 passes the bare `-2`: DSPSR's
                                * static table); all four 0 = the default -3.3359, -1, +1, +3.3359.  A run-time table in
                                * every kernel, so another level scheme is a data change.                                  */
  uint32_t unpack_mode;        /* 0 = the static table above (what this build takes the bare `-2` of process_vdif.py:157,160 to
                                * mean).  1 = DYNAMIC LEVEL SETTING after Jenet & Anderson (1998): per window of dls_nsample
                                * consecutive samples of one polarisation the number of low-state samples estimates the undigitised
                                * power, and the window's two output levels are the power-conserving ones for that estimate (formulas:
                                * DESIGN.md section 2a, restated in oracle/frb_oracle.py dls_table).  An OPTION for sites that find
                                * their digifil does this (SURVEY section 7 hard part 2; unpinnable here: DSPSR is absent): 2-bit
                                * input only, through the generic K1 (slower, section 2a).  The digifil shim selects it with
                                * `-2n<nsample>`, `-2c<cutoff>` or `-2t<threshold>` (DSPSR's spelling of the unpacker options).       */
  uint32_t dls_nsample;        /* window length in samples; 0 = 512.  A power of two in 16..8192 that divides the block length   */
  float dls_cutoff_sigma;      /* windows whose low-state count lies further than this many standard deviations from the count a
                                * Gaussian signal at the nominal threshold gives are zeroed (impulsive interference); 0 = 10; < 0 = off */
  float dls_threshold;         /* sampler threshold in units of the nominal rms; 0 = 0.9674 (the optimum for four levels)        */
} frbch_config;

typedef struct frbch_handle frbch_handle;

/* geometry/result info, valid after frbch_open (time fields after the first frame was seen) */
typedef struct frbch_info {
  uint32_t size;
  uint32_t nchan, freq_res, tscrunch, nif;
  uint64_t block_samples;      /* N = 2*nchan*freq_res real samples per pol per block          */
  uint64_t block_payload_bytes;/* N/2                                                          */
  uint64_t rows_per_block;     /* freq_res / tscrunch output time samples per block            */
  uint64_t row_bytes;          /* nif*nchan*|nbit|/8                                           */
  uint64_t rescale_interval_rows; /* 0 = rescale disabled                                      */
  uint64_t rows_out;           /* output time samples produced so far                          */
  uint64_t blocks_done;
  double tsamp_s;              /* nchan*tscrunch/|bw| us (create_config.py:561)                */
  double tstart_mjd;
  double fch1_mhz, foff_mhz;
  uint32_t frame_bytes, header_bytes;
  uint32_t have_rescale;       /* offset/scale are defined                                     */
  uint32_t diag;               /* diagnostics of the last call: bit 0 = the whole-file path wrote a regular output file through its
                                  preallocated shared mapping (parallel copies) instead of write() calls; bit 1 = a FIFO output
                                  took the rows by reference (vmsplice of the pinned ring; FRBCH_FIFO_COPY=1 forces write())   */
  uint64_t frames_seen;        /* host streaming path: frames whose header was checked           */
  uint64_t frames_invalid;     /* ... with the VDIF invalid bit set: their samples enter the filterbank as 0 (the level table's
                                  mean), extract_baseband_chunk.py:56-69 reads the same bit                            */
  uint64_t frame_gaps;         /* ... frame-number discontinuities; forward jumps are filled with zero samples so that the
                                  stream stays contiguous in time (-cont, process_vdif.py:157), see frames_filled       */
  uint64_t block_stride_bytes; /* payload bytes between block starts: = block_payload_bytes, less with -F C:D */
  uint32_t nfilt_pos, nfilt_neg; /* -F C:D overlap-save: channel samples dropped at the start / end of a block */
  uint64_t frames_filled;      /* zero frames inserted for missing frame numbers (host paths)                              */
} frbch_info;

/* per-kernel device time accumulated since the last frbch_timing_reset (HIP events recorded on
 * the stream the kernels are launched on); enabled by frbch_set_profiling(h, 1). */
typedef struct frbch_timing {
  uint32_t size;
  uint32_t nkernels;
  struct {
    char name[48];
    uint64_t launches;
    double total_ms;
    double algorithmic_bytes;  /* sum over launches of the DESIGN.md per-launch byte model      */
  } k[12];                     /* nkernels of them are in use                                    */
} frbch_timing;

/* ---- configuration helpers ------------------------------------------------------------- */
/* defaults = digifil's as the reference relies on them (nbit 8, -d1, 10 s rescale interval) */
int frbch_config_init(frbch_config* cfg);
/* parse the ASCII side file make_hdr writes (process_vdif.py:115-139, keys :122-133) */
int frbch_config_from_hdr(const char* hdr_path, frbch_config* cfg);
/* parse the digifil argv run_digifil builds (process_vdif.py:156-182):
 * -cont -c -b<n> -S<s> -T<s> -2 -D <dm> [-t <T>] -o <out> <hdr> -threads <n>
 * (-P<p> | -d<n>) -F<C>:<R|D> [-I<secs>]; -D and -F may repeat, last wins.  argv[0] is skipped.
 * hdr_path/out_path receive the positional .hdr and the -o value. */
int frbch_parse_digifil_argv(int argc, const char* const* argv, frbch_config* cfg,
                             char* hdr_path, size_t hdr_cap, char* out_path, size_t out_cap,
                             char* err, size_t err_cap);

/* ---- lifecycle --------------------------------------------------------------------------- */
int frbch_open(const frbch_config* cfg, frbch_handle** out);   /* replaces: digifil start-up  */
void frbch_close(frbch_handle* h);
const char* frbch_last_error(frbch_handle* h);                 /* replaces: digifil's stderr  */
const char* frbch_strerror(int code);
int frbch_get_info(frbch_handle* h, frbch_info* info);
/* forget stream + rescale state (as if freshly opened; buffers and tables are kept) */
int frbch_reset(frbch_handle* h);

/* ---- whole-file path: what `digifil ... -o <out> <hdr>` does (process_vdif.py:191) ---------
 * Reads cfg.datafile-style VDIF at `vdif_path`, honours -S/-T, writes SIGPROC header + samples to
 * `out_fil` strictly sequentially.  out_fil may be an existing FIFO: opened
 * O_WRONLY|O_CREAT|O_TRUNC without O_EXCL (INSTALL.md:32-35), never unlinked, never seeked. */
int frbch_run_file(frbch_handle* h, const char* vdif_path, const char* out_fil);

/* ---- one scan, several IFs on one GPU -------------------------------------
 * What base2fil.sh does with N digifil processes, N FIFOs and sigproc `splice`
 * (base2fil.sh:348-350,404-448), in one call: `ifs` are freshly opened handles on the same device with the same
 * nchan / tscrunch / nbit / products, listed like base2fil's splice_list: highest IF first (:350,367);
 * vdif_paths[i] is the per-IF VDIF of ifs[i].  Every IF's rows are copied into its columns of one pitched device
 * buffer (the frequency concatenation happens in HBM), and ONE SIGPROC file is written -- the
 * <exp>_<st>_no0<scan>_IFall_vdif_pol<pol>.fil of base2fil.sh:389: nchans = nif*nchan, fch1 of ifs[0], rows cut
 * to the shortest IF as splice does.  out_fil may be a FIFO (same open flags as frbch_run_file).
 * Errors are reported through frbch_last_error(ifs[0]). */
int frbch_run_scan(frbch_handle* const* ifs, uint32_t nif, const char* const* vdif_paths, const char* out_fil);

/* The same scan with everything resident in HBM (the device-side form of frbch_run_scan): d_frames[i] holds
 * the per-IF VDIF frames of ifs[i] (same frame geometry and length for all), `nblocks` filterbank blocks of every IF are
 * transformed starting `payload_byte_offset` bytes into the payload streams, and the rows land in ONE row buffer
 * d_rows[row][product][IF-major channels] -- ifs[0] (the highest IF, base2fil.sh:350,367) in the first nchan columns of every
 * (row, product) line: the frequency concatenation `splice` does on the host (base2fil.sh:422) happens in the store
 * addresses of the last kernel.  row_pitch_bytes = nif * (row_bytes of one IF); rows_cap rows fit in d_rows.  With `flush`
 * the pending rescale interval of every IF is closed too (the end of the scan).  *rows_written = rows every IF delivered.
 * The IFs go through the two lanes of DESIGN.md section 4b one behind the other: the front half of IF i + 1 overlaps the
 * back half of IF i.  `stream` as in frbch_process_device.  Errors are reported through frbch_last_error(ifs[0]). */
int frbch_scan_device(frbch_handle* const* ifs, uint32_t nif, const void* const* d_frames, size_t nframes,
                      uint32_t frame_bytes, uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nblocks,
                      int flush, void* d_rows, size_t row_pitch_bytes, uint64_t rows_cap, uint64_t* rows_written, void* stream);

/* ---- streaming host path ----------------------------------------------------------------- */
/* push whole or partial frames (byte stream starting at a frame boundary on the first call) */
int frbch_push(frbch_handle* h, const uint8_t* frames, size_t nbytes);
/* signal end of input: flushes a pending rescale interval */
int frbch_flush(frbch_handle* h);
/* copy out quantised [t][nif][chan] rows that are ready; returns bytes written, <0 on error */
long frbch_pull(frbch_handle* h, uint8_t* dst, size_t cap);
/* SIGPROC header bytes for the stream (valid once the first frame has been seen) */
long frbch_sigproc_header(frbch_handle* h, uint8_t* dst, size_t cap);

/* ---- device-resident path (inputs and outputs already in HBM) ----------------------------- */
/* d_frames: device pointer to whole VDIF frames (frame geometry taken from `frame_bytes`,
 * `header_bytes`); the payload stream is entered `payload_byte_offset` bytes after the first
 * payload byte (any value; the fast gather needs it and the payload size to be multiples of the
 * per-row piece, 2..16 bytes, and d_frames 16-byte aligned, else the generic kernel is used);
 * `nblocks` filterbank blocks are transformed.  d_out / d_power must be 16-byte aligned.
 * d_out receives rows_per_block*nblocks rows of row_bytes (fewer while a rescale interval is still
 * being measured; *rows_written says how many).  `stream` is a hipStream_t (NULL = the handle's
 * own stream).  Asynchronous with respect to the host except when a rescale interval completes. */
int frbch_process_device(frbch_handle* h, const void* d_frames, size_t nframes,
                         uint32_t frame_bytes, uint32_t header_bytes, uint64_t payload_byte_offset,
                         uint64_t nblocks, void* d_out, size_t out_cap_bytes,
                         uint64_t* rows_written, void* stream);
/* finish a pending rescale interval into d_out (device) */
int frbch_flush_device(frbch_handle* h, void* d_out, size_t out_cap_bytes, uint64_t* rows_written,
                       void* stream);
/* detected + scrunched power of `nblocks` blocks as float32 [t][nif][chan] (channel order of the
 * output file), no rescale/digitise: the "-b-32 -I0" data product and the parity-test tap. */
int frbch_power_device(frbch_handle* h, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                       uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nblocks,
                       float* d_power, size_t cap_bytes, void* stream);

/* The unpack stage (A4) in isolation: `nsamples` dual-pol samples starting `payload_byte_offset` bytes into the payload
 * stream, decoded with the handle's level table (frbch_config::levels) exactly as the filterbank's first kernel decodes
 * them: float32 d_volt[pol][nsamples].  decoder 0 = the generic K1's decode (1- and 2-bit input); 1 = the register
 * kernels' two decodes (nibble table of frbch_k1_wave, select chain of frbch_k1_fast; 2-bit input, nsamples even):
 * d_volt[2][pol][nsamples].  The `-2` of process_vdif.py:157,160 selects this static table. */
int frbch_unpack_device(frbch_handle* h, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                        uint32_t header_bytes, uint64_t payload_byte_offset, uint64_t nsamples, int decoder,
                        float* d_volt, size_t cap_bytes, void* stream);

/* ---- rescale state (the one stateful stage; SURVEY 7 hard part 6) ------------------------- */
/* offset/scale are [nif][nchan] in INPUT channel order k (ascending FFT bin), float32 */
int frbch_get_rescale(frbch_handle* h, float* offset, float* scale);
int frbch_set_rescale(frbch_handle* h, const float* offset, const float* scale);

/* ---- after the filterbank ------------------------------------------------
 * The rows of a SIGPROC filterbank -- [t][nifs][nchan] samples of 8 or 16 bit (unsigned) or float32, as frbch_run_file /
 * frbch_run_scan write them -- dedispersed incoherently or folded on the GPU.  The reference delegates both to external
 * programs: PRESTO's prepdata / prepsubband (process_vdif.py:202-229: `-dm`, `-lodm -numdms -dmstep`, `-zerodm`,
 * `-clip`, always `-nobary -noweights -noscales`) and `dspsr -E <par> -L 10 -A -d1 <IFall.fil>` (base2fil.sh:474).
 * Neither program is in the reference tree: the conventions are this library's (oracle/post_oracle.py states them).
 * `_host` variants take host pointers and move the data themselves; `_device` variants work on rows already in HBM. */
typedef struct frbch_fil_desc {
  uint32_t size;               /* = sizeof(frbch_fil_desc)                                                      */
  uint32_t nchan, nifs;        /* SIGPROC nchans, nifs                                                           */
  int32_t nbits;               /* 8, 16 (unsigned codes) or 32 (float32)                                         */
  uint32_t product;            /* which of the nifs products is used (0 = PP+QQ of -d1, or I of IQUV)            */
  uint32_t reserved;
  double fch1_mhz, foff_mhz;   /* SIGPROC fch1 / foff (centre of channel 0, channel step; foff < 0 as written here) */
  double tsamp_s, tstart_mjd;
} frbch_fil_desc;

/* Output samples per DM: nrows minus the largest delay of any requested DM (delay of channel c = DM / 2.41e-4 *
 * (f_c^-2 - f_top^-2) s, rounded to samples as int(x + 0.5)).  < 0: bad arguments. */
long frbch_dedisperse_nout(const frbch_fil_desc* fil, uint64_t nrows, const double* dms, uint32_t ndm);
/* out[dm][t] = sum_c x[t + delay_c(dm)][c]  (float32), after the optional time-domain clip (`-clip <sigma>`: a time
 * sample whose zero-DM sum lies more than clip_sigma sigma off the mean -- two rounds -- is replaced by the channel means
 * of the unclipped samples) and the optional zero-DM filter (`-zerodm`: the mean over channels of every time sample is
 * subtracted).  *nclipped receives the number of clipped time samples.  Integer rows: every sum is exact. */
int frbch_dedisperse_host(const frbch_fil_desc* fil, const void* rows, uint64_t nrows, const double* dms, uint32_t ndm,
                          uint32_t zerodm, double clip_sigma, int device, float* out, uint64_t nout,
                          uint64_t* nclipped, char* err, size_t err_cap);
int frbch_dedisperse_device(const frbch_fil_desc* fil, const void* d_rows, uint64_t nrows, const double* dms, uint32_t ndm,
                            uint32_t zerodm, double clip_sigma, int device, float* d_out, uint64_t nout,
                            uint64_t* nclipped, char* err, size_t err_cap);
/* Sub-integrations of subint_s seconds (dspsr -L): ceil(nrows / round(subint_s / tsamp)). */
long frbch_fold_nsub(const frbch_fil_desc* fil, uint64_t nrows, double subint_s);
/* Phase fold with the spin of a .par file: turns(tau) = F0 tau + F1 tau^2 / 2, tau = (tstart - PEPOCH) + t tsamp
 * [- delay_c(dm) when apply_delays: incoherent inter-channel dedispersion at fold time; 0 = what dspsr does with a
 * filterbank: channels are folded as they arrive and the archive keeps DM for a later `dedisperse`].  Topocentric: no
 * barycentric, binary or position terms.  profile[sub][bin][chan] = sum of the samples, hits[...] = their number. */
int frbch_fold_host(const frbch_fil_desc* fil, const void* rows, uint64_t nrows, double f0_hz, double f1, double pepoch_mjd,
                    double dm, uint32_t apply_delays, uint32_t nbin, double subint_s, int device, double* profile,
                    uint32_t* hits, uint32_t nsub, char* err, size_t err_cap);
int frbch_fold_device(const frbch_fil_desc* fil, const void* d_rows, uint64_t nrows, double f0_hz, double f1,
                      double pepoch_mjd, double dm, uint32_t apply_delays, uint32_t nbin, double subint_s, int device,
                      double* d_profile, uint32_t* d_hits, uint32_t nsub, char* err, size_t err_cap);

/* ---- in front of the filterbank: the corner turn -------------------------------
 * jive5ab's spif2file splits the recorder's stream -- every W-bit word holds one time sample of ALL channels -- into one
 * 2-channel stream per IF, driven by the recipe strings of spif2file.sh:31-113, e.g. the 16-channel 2-bit mode
 * `32>[24,25,16,17][8,9,0,1]...[14,15,6,7]:0-7`: output stream ("tag") g takes the listed bits of every word, in that
 * order, LSB first; `swap_sign_mag+` (Mark5B modes, :79-94) first exchanges the two bits of every 2-bit sample.  Doing
 * it on the GPU lets frbch_process_device read the result straight from HBM (header_bytes = 0) without per-IF files.
 * jive5ab is not in the reference tree: the bit order is this library's restatement (oracle/post_oracle.py). */
int frbch_cornerturn_info(const char* recipe, uint32_t* word_bits, uint32_t* ntags, uint32_t* bits_per_word,
                          uint32_t* first_tag, char* err, size_t err_cap);
/* frames: nframes recorder frames; out[g]: payload bytes of tag first_tag + g, out_bytes_each = words * bits_per_word / 8 */
int frbch_cornerturn_host(const char* recipe, const void* frames, size_t nframes, uint32_t frame_bytes,
                          uint32_t header_bytes, void* const* out, uint32_t ntags, size_t out_bytes_each, int device,
                          char* err, size_t err_cap);
int frbch_cornerturn_device(const char* recipe, const void* d_frames, size_t nframes, uint32_t frame_bytes,
                            uint32_t header_bytes, void* const* d_out, uint32_t ntags, size_t out_bytes_each, int device,
                            char* err, size_t err_cap);

/* ---- measurement -------------------------------------------------------------------------- */
int frbch_set_profiling(frbch_handle* h, int enable);
int frbch_timing_reset(frbch_handle* h);
int frbch_get_timing(frbch_handle* h, frbch_timing* t);

/* library self-description: "frbch <abi> gfx950 ..." */
const char* frbch_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FRBCH_H */
