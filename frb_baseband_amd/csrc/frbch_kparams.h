// Kernel parameter block shared by the host side of the C-ABI library and the HIP kernels.
#ifndef FRBCH_KPARAMS_H
#define FRBCH_KPARAMS_H
#include <stdint.h>

struct __attribute__((aligned(8))) cf {  // complex float; 8-byte aligned so LDS traffic is b64
  float x, y;
};

struct __attribute__((aligned(16))) f4 {  // 16-byte vector of floats for coalesced row traffic
  float v[4];
};

enum { FRBCH_OUT_FLOAT_POWER = 0, FRBCH_OUT_CODES = 1,
       FRBCH_OUT_STATS = 2 };   // rescale sums only, no rows (frbch_k2_priv: first pass of the two-pass rescale)

struct KParams {
  // ---- geometry -------------------------------------------------------------------------
  int log2_c2;        // log2(2C): length of the across-branch FFT
  int log2_r;         // log2(R):  length of the along-branch FFTs
  int c;              // nchan
  int c2;             // 2C  = number of polyphase branches
  int r;              // freq_res
  int g;              // branches per K1 workgroup (power of two, divides 2C)
  int tt;             // time samples per K2 sub-tile (power of two, divides R)
  int tscr;           // tscrunch T (power of two, divides R)
  int nif;            // 1 or 4 products
  int pol_mode;       // 0,1,2,3,4
  int nbit;           // 2, 8, 16, -32
  int flip;           // 1 = USB input: reverse channel order on output
  int out_mode;       // FRBCH_OUT_FLOAT_POWER / FRBCH_OUT_CODES
  int log2_nlo;       // split of the N-point twiddle table: q = hi << log2_nlo | lo
  int log2_g;         // log2(g)
  int in_bits;        // bits per input sample: 2 or 1
  // ---- VDIF frame stream ------------------------------------------------------------------
  uint32_t frame_bytes, header_bytes, payload_bytes;
  uint32_t rel0;            // wave kernels: payload byte (inside frame 0 of `frames`) of block 0
  uint64_t payload_off;     // payload byte at which block 0 of this launch starts
  const uint8_t* frames;
  // ---- work buffers -------------------------------------------------------------------------
  uint64_t gs;              // group stride of the spill in cf: R*g + pad (pad keeps the 2C/g slabs off one HBM channel)
  cf* spill;                // [nblk][2C/g][gs >= R*g]: rows [t][g] of the delayed branch series w[n1][t]
  cf* s_dc;                 // [nblk][2C]           S[n1] = sum_n2 p[n1 + 2C n2]
  cf* p0;                   // [nblk][2C]           dP[k'] = P[(k'+1)R] - P[k'R], P[k'R] = FFT_2C(S)
  const cf* tw_r;           // exp(-2 pi i k / R),  k < R/2
  const cf* tw_c2;          // exp(-2 pi i k / 2C), k < C
  const cf* tw_nhi;         // exp(-2 pi i (h << log2_nlo) / N)
  const cf* tw_nlo;         // exp(-2 pi i l / N)
  // fast-path tables (kernels_fast.inc): L = 256*M, TPS = 16*M
  const cf* ftw1_r;         // [16][R/16]   exp(-2 pi i p ka / R)
  const cf* ftw2_r;         // [M][16]      exp(-2 pi i c kb / (R/16))
  const cf* ftw1_h;         // same two tables for length R/2 (frbch_k1_split: half transforms)
  const cf* ftw2_h;
  const cf* ftw1_c;         // same for the across-branch length 2C
  const cf* ftw2_c;
  const cf* td1;            // [2C][16]     exp(-2 pi i n1 kc / (16*2C))      (delay, coarse part)
  const cf* td2;            // [2C][R/16]   exp(-2 pi i n1 k0 / N)            (delay, fine part)
  const float* offset;      // [nif][C], input channel order k
  const float* scale;       // [nif][C]
  float* power_out;         // [row][nif][C] float32, output channel order
  uint8_t* code_out;        // [row][nif][C] packed to nbit, output channel order
  uint64_t row0;            // first output row of block 0 of this launch inside *_out
  uint64_t out_pitch;       // code_out only: values between the starts of consecutive (row, product) lines; = C for packed rows,
                            // larger when the rows land in this IF's columns of a scan's row buffer (frbch_scan_device / frbch_run_scan)
  uint32_t div_magic, div_shift;  // x / payload_bytes = (t + ((x - t) >> 1)) >> div_shift, t = mulhi(magic, x)
  float lut[4];             // 2-bit level table
  float digi_mean, digi_scale, digi_max;
  cf rot6[6];               // K1 remapped pass: exp(-2 pi i * step * k / R), k = 1,2,3,4,8,12, step = 64/branches per WG
  // ---- coherent dedispersion (-F C:D): K1 forward only -> K2c chirp -> K3 inverse+detect -> K4 transpose
  int coherent;             // 1 = that pipeline
  int nfilt_pos;            // channel samples discarded at the start of every block (overlap-save)
  int keep;                 // channel samples kept per block, a multiple of tscr
  int stag;                 // K1 wave kernels: priority schedule of the two halves of a workgroup (bit mask, see the kernel)
  uint64_t hop;             // real samples between block starts: N, or 2C*keep with overlap-save
  cf* spill2;               // [nblk][2C/4][R][4]  chirped spectrum P'[k'][pos]: groups of 4 channels side by side, so that a
                            // K2c tile (few positions x all channels) writes 32-byte x tile-width runs; pos = position of
                            // fine bin j in the order the along-branch transform leaves it (S2_INDEX below)
  const cf* chirp;          // [2C/4][R][4]   Hermitian-extended dedispersion kernel, same order and layout
  float* ptmp;              // [nblk][nif][C][keep/T]  detected + scrunched power, channel-major
  const float* scr_in;      // two-stage tscrunch (2C = 8192, tscrunch > 2): the wave K2's rows of two time samples each, [row][C] floats
  uint32_t scr_fact;        // ... rows of scr_in per output row (tscrunch / 2)
  uint64_t scr_rows;        // ... output rows of this launch
  double* stat_partial;     // fast K2, float power: [workgroup row][ncol][2] running (sum, sum of squares); null = off
  uint64_t stat_limit;      // ... of the rows below this absolute row of power_out (the end of the rescale interval)
  uint32_t nblk;            // blocks in this launch (persistent kernels loop over them)
  uint32_t dbg;             // timing-only ablations (cfg.flags >> 8); results are wrong when set
  const uint8_t* stg;        // wave K1: the launch's payload re-ordered by frbch_k0_stage, [blk][branch group][row][RB bytes]: the
                             // rows of one workgroup and block are contiguous (null = gather from the frames)
  uint8_t* stg_out;          // frbch_k0_stage: where it writes that buffer
  int tile_major;            // 0 = slab layout.  2 = this launch's spill is [blk][t/2][n1/G][t%2][n1%G] (paired-branch wave K1 -> wave K2
                             // with two time samples per workgroup: a K2 tile is one contiguous run).  8 = chunks of eight time samples
                             // [blk][t/8][n1/G][t%8][n1%G] (frbch_k1_fast<5> -> frbch_k2_fast<5>: whole cache lines on both sides).
                             // Set per launch by the engine (the K1 that runs decides)
  unsigned long long* stamps; // diagnostic builds of the wave K1: s_memtime stamps [workgroup][wave][16] of one block; null = off
  const uint32_t* fbad;       // generic K1 only: bitmap over the frames of `frames` (bit f = frame f is flagged invalid or is a
                              // filler for a missing frame): its samples enter the filterbank as 0.  null = every frame is good
  uint64_t fbad_frame0;       // index, inside the bitmap, of the frame `frames` points at
  // ---- dynamic level setting (frbch_config.unpack_mode 1; generic K1 and the unpack tap) ---------------------------------------
  uint32_t* dls_nlow;         // [window] low-state counts of the windows of this launch: pol 0 | pol 1 << 16 (frbch_dls_count writes,
                              // the unpack reads); window w = samples [w << dls_lg_ns, (w + 1) << dls_lg_ns) counted from the launch's
                              // first sample (block starts are multiples of the window: frbch_host.cpp checks)
  const float* dls_tab;       // [nsample + 1][2]: output levels (low, high) for a window with that many low-state samples; (0, 0) for
                              // counts outside the accepted range (the window is zeroed).  null = the static table `lut`
  int dls_lg_ns;              // log2(window length in samples)
};

struct StatParams {
  const float* power;       // [rows][ncol] (ncol = nif*C, output channel order)
  double* partial;          // [nchunk][ncol][2]
  uint64_t rows;            // rows that enter the statistics
  uint64_t rows_per_chunk;
  int ncol, c, nif, flip, nchunk;
  int rsplit;               // frbch_stats_partial: threads sharing one column group (rows interleaved), > 1 for narrow rows; the
                            // final reduction then runs over nchunk = chunks x rsplit rows of partial sums
  int cpw;                  // frbch_stats_final: columns per workgroup (8 = whole lines of the partial sums; 2 when there are few columns)
  float* offset;            // [nif][C] input channel order
  float* scale;
};

struct QuantParams {
  const float* power;       // [rows][ncol]
  uint8_t* out;
  uint64_t rows;
  int ncol, c, nif, flip, nbit;
  const float* offset;
  const float* scale;
  float digi_mean, digi_scale, digi_max;
  uint32_t grid_x;          // workgroups launched (grid-stride loop)
  int log2_c;
  int log2_ncol;
  uint64_t pitch;           // output values between consecutive (row, product) lines (KParams::out_pitch)
  uint32_t rphases;         // frbch_quantise_fast: row phases = grid_x * 256 / (ncol / 4); a thread takes rows phase + i * rphases
};
#define QUANT_GRID_X(p) ((p).grid_x)

struct ChirpParams {        // frbch_chirp_build: fills KParams::chirp once per handle
  cf* chirp;
  int c, c2, r, log2_r;
  int usb;                  // 1 = BW > 0
  int order_m;              // 0: fine bin j sits at its bit-reversed position (generic K1); M > 0: at the position the register
                            // passes radix 16 x M x 16 leave it: pos = kc*16M + ka*M + kb for j = ka + 16 kb + 16M kc
  double band_edge_mhz;     // sky frequency of baseband 0: lower band edge (USB) / upper band edge (LSB)
  double df_mhz;            // channel width |BW|/C
  double dm_over_k;         // DM / 2.41e-4  [s MHz^2]
};

// element (channel row, position) of one block of spill2 / of the chirp table; r = freq_res
// value index of (row, product, first channel) in code_out: lines of out_pitch values per (row, product)
#define FRBCH_CODE_INDEX(pitch, row, nif, prod, cout) ((((uint64_t)(row)) * (uint64_t)(nif) + (uint64_t)(prod)) * (uint64_t)(pitch) + (uint64_t)(cout))

#define S2_INDEX(row, pos, r) (((((uint64_t)(row) >> 2) * (uint64_t)(r) + (uint64_t)(pos)) << 2) + (uint64_t)((row) & 3))

#endif
