// Host-side (no GPU) pieces of libfrbch: formats, configuration, geometry planning.
#ifndef FRBCH_HOST_H
#define FRBCH_HOST_H
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/frbch.h"

namespace frbch {

struct VdifInfo {
  uint32_t invalid, legacy, seconds, ref_epoch, frame_nr, log2_nchan, frame_bytes, is_complex,
      bits_per_sample, thread_id, station_id;
  uint32_t header_bytes() const { return legacy ? 16u : 32u; }
  uint32_t payload_bytes() const { return frame_bytes - header_bytes(); }
};

// first 16 bytes of a frame -> fields (VDIF 1.1; geometry consumed at base2fil.sh:395-401)
bool parse_vdif_header(const uint8_t* buf, VdifInfo* out);
// checks that the stream is what spif2file writes for one IF (spif2file.sh:178-186)
bool check_vdif_supported(const VdifInfo& v, std::string* why);
int vdif_epoch_mjd(int ref_epoch);

// geometry of one handle, derived from the configuration
struct Plan {
  int c, r, c2, log2_c2, log2_r, log2_n, log2_nlo;
  int g, tt, tscr, nif, flip, nthreads;
  int in_bits;                // bits per input sample (2, or 1)
  uint64_t n;                 // samples per pol per block
  uint64_t gs;                    // spill group stride in cf (R*g + pad)
  uint64_t block_payload_bytes;   // payload bytes one block reads (N samples)
  uint64_t block_stride_bytes;    // payload bytes between block starts (= block_payload_bytes unless coherent)
  uint64_t hop;                   // the same in samples: N, or 2C*keep
  int coherent;                   // -F C:D pipeline
  int nfilt_pos, nfilt_neg, keep; // overlap-save: discarded at start / end, kept (multiple of tscr); keep = R otherwise
  size_t k3_lds, k4_lds;
  uint64_t rows_per_block;
  uint64_t row_bytes;         // bytes per output row
  uint64_t ncol;              // nif * C
  uint64_t interval_rows;     // rescale interval in output rows; 0 = disabled
  size_t k1_lds, k2_lds, kc_lds;
  uint32_t maxb;              // blocks per launch
  int fast_k1_log2m;          // 0 = generic K1, else log2(M) with R = 256*M
  int fast_k2_log2m;          // 0 = generic K2 (unless fast_k2_m1), else log2(M) with 2C = 256*M
  int spill_tile_major;       // 2 / 8 = the kernel pair in use supports that tile-major spill (KParams::tile_major), 0 = slab layout
  int fast_k2_m1;             // 1 = wave-private K2 for 2C = 256 (M = 1; Kc stays generic)
  int coh_nt;                 // threads per K3 workgroup: 1024, or 512 at R = 4096 (74 KB of LDS: two workgroups per CU)
  int fast_k2_nt;             // threads per K2 workgroup (512 / 1024)
  size_t k1_fast_lds, k2_fast_lds;
  int fast_k1_split;          // 1 = frbch_k1_split (R = 2048, staged input): 16 independent waves per CU; falls back to the wave K1
  size_t k1_split_lds;
  int fast_k1_g;              // branches per wave-private K1 workgroup (<= g)
  int fast_k1_kind;           // M = 8 only: 0 = 8 waves x 8 branches, 1 = 4 waves x 4 branches, 2 = 8 waves x 4 branches (2 waves/seq)
  int fast_k2_nw;             // waves per wave-private K2 workgroup (2 or 4)
  int fast_k1_wave, fast_k2_wave; // 1 = wave-private variant (8 / 4 waves per workgroup), 0 = barrier variant
  int fast_k2_priv;           // 2C = 2048 behind the paired-branch wave K1 (tile-major spill): frbch_k2_priv, one wave per time sample, no barrier
                              // inside the transform (kernels_k2priv.inc); launches whose K1 fell back to another layout keep frbch_k2_wave
  size_t k2_priv_lds;
  int fast_k2_lane;           // 2C = 64 / 128 (nchan 32 / 64): frbch_k2_lane, a whole across-branch sequence per lane (1) or lane pair (2); 0 = off
  int k2_two_stage;           // tscrunch beyond the wave K2's tile: K2 writes float rows of k2_stage1_tscr time samples, frbch_k2_scrunch sums k2_two_stage of them (0 = off)
  int k2_stage1_tscr;         // ... the tscrunch K2 itself runs with (2 at 2C = 8192, else its 8-sequence tile)
  // coherent pipeline (-F C:D) on the register-pass kernels: log2(M) of R (K1 forward-only + K3) and of 2C (K2c); 0 = generic
  int coh_fast_r, coh_fast_c;
  size_t k2c_fast_lds, k3_fast_lds;
  double rate_in;             // real samples / s / pol
  double rate_out;            // output rows / s
  double tsamp_s;
  double fch1, foff;
  float digi_mean, digi_scale, digi_max;
  int dls_lg_ns;              // dynamic level setting (cfg.unpack_mode 1): log2 of the window length in samples; 0 = static table
};

// returns "" on success, else the reason (InputError territory)
std::string make_plan(const frbch_config& cfg, Plan* plan, size_t lds_limit, int in_bits = 0);

// overlap-save geometry of the coherent filterbank; "" on success
std::string coherent_geometry(double freq_mhz, double bw_mhz, uint32_t nchan, uint32_t freq_res, uint32_t tscrunch,
                              double dm, uint32_t* r, int* nfilt_pos, int* nfilt_neg);
constexpr double kDmDispersion = 2.41e-4;   // DSPSR's constant: delay = DM / (2.41e-4 nu_MHz^2) s
constexpr uint32_t kMaxCoherentFreqRes = 8192;

std::vector<uint8_t> sigproc_header(const frbch_config& cfg, const Plan& plan, double tstart_mjd, int nchans_total = 0);
double sigproc_angle(const char* text);
int sigproc_telescope_id(const char* name);

// dynamic level setting: output levels (low, high) per low-state count 0..nsample of a window, [nsample + 1][2] (DESIGN.md
// section 2a; (0, 0) = the window is zeroed); cutoff_sigma < 0 = no excision
std::vector<float> dls_table(uint32_t nsample, float cutoff_sigma, float threshold);

// exp(-2 pi i k / n) for k in [0, count), evaluated in double, stored as float pairs
void fill_twiddles(float* dst_xy, uint64_t n, uint64_t count, uint64_t step);

}  // namespace frbch
#endif
