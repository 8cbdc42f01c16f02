"""CPU oracle (numpy, fp64) for the VDIF -> SIGPROC-filterbank channeliser path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or executed by the
product path (``frb_baseband_amd``); only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may use it, and only as the checker.

PARITY UNPINNED (numerical stages).  The reference (pharaofranz/frb-baseband) contains no
numerical code: the arithmetic lives in DSPSR's ``digifil``, which the reference launches as a
subprocess (process_vdif.py:157-191).  DSPSR is an un-vendored, un-pinned third-party dependency
(README.md:8; fallback commit b68528e15e8, INSTALL.md:37-39), absent from /root/reference and from
this image, and the reference ships no tests or golden outputs for it.  This file therefore
restates digifil's *published* stage chain as selected by the flags the reference passes
(process_vdif.py:156-182) and anchors on the reference's own call sites:

  stage                      reference anchor (file:line)                      function here
  -------------------------  -----------------------------------------------   ---------------------
  .hdr ASCII side file       process_vdif.py:115-139 (keys :122-133)           read_hdr
  VDIF frame geometry        base2fil.sh:130-147,395-401; spif2file.sh:181     parse_vdif_header, strip_frames
  2-bit unpack  (-2)         process_vdif.py:157,160; spif2file.sh:34          unpack_2bit
  filterbank    (-F C:R)     process_vdif.py:162-171                           filterbank_block
  detection     (-d/-P)      process_vdif.py:58-64,163-176; base2fil.sh:214-7  detect
  time scrunch  (-t)         process_vdif.py:156-158                           tscrunch
  rescale       (-c, -I0)    process_vdif.py:157,160,181-182; frb.conf:51      rescale_stats, rescale_apply
  digitise      (-b)         process_vdif.py:65-68,153-155                     digitise
  SIGPROC output (-o)        process_vdif.py:143-145; base2fil.sh:422          sigproc_header, channelise
  resolution identity        create_config.py:561                               (checked in tests)

The harness part of the path (make_hdr / run_digifil argv mapping) IS pinned: golden strings are
captured by importing the reference's process_vdif.py (tests/golden/make_harness_golden.py).

Conventions fixed here (each is a statement of DSPSR's documented behaviour, unverifiable in this
image, see DESIGN.md "Oracle"):
  * real-sampled input, N = 2*C*R samples per pol per block; forward real FFT (unnormalised);
    channel k owns bins [k*R, (k+1)*R) (Nyquist bin dropped); backward complex FFT of length R
    (unnormalised) gives R time samples of channel k; no overlap when not dedispersing.
  * only whole blocks are transformed; the tail that does not fill a block is dropped.
  * 2-bit offset-binary levels 0..3 -> (-3.3359, -1, +1, +3.3359) (static table).
  * rescale: per (product, channel) offset = -mean, scale = 1/sqrt(variance) (population variance)
    over the first ``interval`` output samples (default 10 s worth, clipped to what exists);
    with -c the pair is then frozen; without -c it is recomputed per interval; -I0 disables.
  * digitise: int(x*scale_n + mean_n + 0.5) clipped to [0, 2^n - 1], with (mean_n, scale_n) =
    (1.5, 1) for 2 bit, (127.5, 127.5/6) for 8 bit, (32767.5, 32767.5/6) for 16 bit; -32 = float32.
  * output order: time-major [t][product][channel], channels in DESCENDING sky frequency
    (foff < 0): USB input (BW > 0) is flipped, LSB input (BW < 0) is already descending.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

LEVELS_2BIT = np.array([-3.3359, -1.0, 1.0, 3.3359], dtype=np.float64)
DIGI_SIGMA = 6.0
DEFAULT_RESCALE_INTERVAL_S = 10.0
MJD_2000 = 51544  # MJD of 2000-01-01, the VDIF epoch-0 reference


# --------------------------------------------------------------------------------------------
# .hdr side file (process_vdif.py:122-133)
# --------------------------------------------------------------------------------------------
def read_hdr(path: str) -> dict:
    """Parse the 12-line DSPSR ASCII header written by make_hdr (process_vdif.py:115-139)."""
    out = {}
    with open(path, "r") as f:
        for line in f.read().splitlines():
            parts = line.split(None, 1)
            if len(parts) == 2:
                out[parts[0]] = parts[1].strip()
    for key in ("FREQ", "BW"):
        out[key] = float(out[key])
    out["NPOL"] = int(out.get("NPOL", 2))
    return out


# --------------------------------------------------------------------------------------------
# VDIF frames (public VDIF 1.1 layout; geometry used at base2fil.sh:395-401)
# --------------------------------------------------------------------------------------------
@dataclass
class VdifHeader:
    invalid: int
    legacy: int
    seconds: int
    ref_epoch: int
    frame_nr: int
    version: int
    log2_nchan: int
    frame_bytes: int
    is_complex: int
    bits_per_sample: int
    thread_id: int
    station_id: int

    @property
    def header_bytes(self) -> int:
        return 16 if self.legacy else 32

    @property
    def payload_bytes(self) -> int:
        return self.frame_bytes - self.header_bytes


def parse_vdif_header(buf: bytes) -> VdifHeader:
    w0, w1, w2, w3 = struct.unpack_from("<4I", buf, 0)
    return VdifHeader(
        invalid=(w0 >> 31) & 1, legacy=(w0 >> 30) & 1, seconds=w0 & 0x3FFFFFFF,
        ref_epoch=(w1 >> 24) & 0x3F, frame_nr=w1 & 0xFFFFFF,
        version=(w2 >> 29) & 0x7, log2_nchan=(w2 >> 24) & 0x1F, frame_bytes=(w2 & 0xFFFFFF) * 8,
        is_complex=(w3 >> 31) & 1, bits_per_sample=((w3 >> 26) & 0x1F) + 1,
        thread_id=(w3 >> 16) & 0x3FF, station_id=w3 & 0xFFFF)


def vdif_epoch_mjd(ref_epoch: int) -> int:
    """MJD of the VDIF reference epoch (6-month steps from 2000-01-01)."""
    year = 2000 + ref_epoch // 2
    month = 1 if ref_epoch % 2 == 0 else 7
    # days from civil (proleptic Gregorian) -> MJD
    a = (14 - month) // 12
    y = year + 4800 - a
    m = month + 12 * a - 3
    jdn = 1 + (153 * m + 2) // 5 + 365 * y + y // 4 - y // 100 + y // 400 - 32045
    return jdn - 2400001


def strip_frames(raw: np.ndarray, frame_bytes: int, header_bytes: int) -> np.ndarray:
    """Drop frame headers; ``raw`` is a whole number of frames (uint8)."""
    nfr = raw.size // frame_bytes
    return raw[: nfr * frame_bytes].reshape(nfr, frame_bytes)[:, header_bytes:].reshape(-1)


def assemble_stream(raw: np.ndarray, rate: float):
    """Frames as they come -> (payload bytes of a stream that is contiguous in time, bad[frame] flags, counters).

    The reference passes `-cont` (process_vdif.py:157,160: treat the input as contiguous) and the tools around it read
    the VDIF invalid flag and frame numbers (extract_baseband_chunk.py:56-69).  Convention of this build (DSPSR absent):
    a frame whose invalid bit is set contributes ZERO voltages (the mean of the level table); a forward jump of the
    frame number (seconds * fps + frame_nr) is filled with as many zero frames, so that later samples keep their time;
    a backward jump is only counted."""
    h0 = parse_vdif_header(raw[:32].tobytes())
    fb, hb = h0.frame_bytes, h0.header_bytes
    nfr = raw.size // fb
    frames = raw[: nfr * fb].reshape(nfr, fb)
    w = frames[:, :8].copy().view("<u4")                       # words 0, 1 of every header
    invalid = ((w[:, 0] >> 31) & 1).astype(bool)
    fps = int(round(rate * 2 * h0.bits_per_sample / 8 / h0.payload_bytes))
    idx = (w[:, 0] & 0x3FFFFFFF).astype(np.int64) * fps + (w[:, 1] & 0xFFFFFF).astype(np.int64)
    pieces, bad = [], []
    gaps = filled = 0
    nxt = None
    for f in range(nfr):
        if nxt is not None and idx[f] != nxt:
            gaps += 1
            if idx[f] > nxt:
                nfill = int(idx[f] - nxt)
                pieces.append(np.zeros(nfill * (fb - hb), np.uint8))
                bad += [True] * nfill
                filled += nfill
        nxt = idx[f] + 1
        pieces.append(frames[f, hb:])
        bad.append(bool(invalid[f]))
    payload = np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)
    return payload, np.array(bad, dtype=bool), dict(gaps=gaps, filled=filled, invalid=int(invalid.sum()))


# --------------------------------------------------------------------------------------------
# 2-bit unpack (A4): byte -> 4 floats
# --------------------------------------------------------------------------------------------
def unpack_2bit(payload: np.ndarray, levels=None) -> np.ndarray:
    """u8[nbytes] -> f64[2][2*nbytes].

    2-channel (= 2 pol), 2-bit VDIF: bits[1:0]=pol0 t, [3:2]=pol1 t, [5:4]=pol0 t+1,
    [7:6]=pol1 t+1 (channel-interleaved, LSB first; recipe brackets at spif2file.sh:34 list
    2 pols x 2 bits per IF).
    """
    b = payload.astype(np.uint8)
    lv = LEVELS_2BIT if levels is None else np.asarray(levels, dtype=np.float32).astype(np.float64)
    out = np.empty((2, b.size * 2), dtype=np.float64)
    out[0, 0::2] = lv[b & 3]
    out[1, 0::2] = lv[(b >> 2) & 3]
    out[0, 1::2] = lv[(b >> 4) & 3]
    out[1, 1::2] = lv[(b >> 6) & 3]
    return out


# --------------------------------------------------------------------------------------------
# 2-bit unpack with DYNAMIC LEVEL SETTING (optional: frbch_config.unpack_mode 1, include/frbch.h)
# --------------------------------------------------------------------------------------------
# SURVEY section 7 (hard part 2) flags that DSPSR's two-bit unpacker may set its output levels per window of the input (Jenet &
# Anderson 1998, PASP 110, 1467) instead of reading the static table above; the reference only passes the bare `-2`
# (process_vdif.py:157,160) and DSPSR is absent, so WHICH it does cannot be settled here.  This is the restatement of the
# dynamic scheme this build offers as an option, so that a site can compare both against a real digifil output.  [EXT-UNVERIFIED]
#   * a window = `nsample` consecutive samples of ONE polarisation, counted from the first sample handed to the filterbank (after -S);
#   * k = samples of the window in a low state (offset binary 1, 2); Phi = k / nsample estimates erf(t / (s sqrt 2)) for a Gaussian
#     voltage of rms s and sampler threshold t, hence u = t / s = sqrt 2 erfinv(Phi) and s = t / u, t = `threshold` nominal rms;
#   * the window's output levels conserve the mean power of the two regions of that Gaussian:
#       lo^2 = s^2 (1 - g / Phi),   hi^2 = s^2 (1 + g / (1 - Phi)),   g = sqrt(2 / pi) u exp(-u^2 / 2)
#     (so Phi lo^2 + (1 - Phi) hi^2 = s^2: the unpacked power follows the undigitised power);
#   * windows with k further than `cutoff_sigma` x sqrt(L Phi0 (1 - Phi0)) from L Phi0, Phi0 = erf(threshold / sqrt 2), and k = 0 or L,
#     are zeroed (impulsive interference excision).
DLS_THRESHOLD = 0.9674
DLS_NSAMPLE = 512
DLS_CUTOFF_SIGMA = 10.0


def dls_table(nsample: int = DLS_NSAMPLE, cutoff_sigma: float = DLS_CUTOFF_SIGMA, threshold: float = DLS_THRESHOLD) -> np.ndarray:
    """f32[nsample + 1][2]: (low, high) output level of a window with k low-state samples; (0, 0) = window zeroed.
    cutoff_sigma < 0: no excision (only k = 0 and k = nsample are zeroed)."""
    from scipy.special import erf, erfinv
    thr = float(threshold)
    phi0 = float(erf(thr / np.sqrt(2.0)))
    mean, sd = nsample * phi0, np.sqrt(nsample * phi0 * (1.0 - phi0))
    kmin, kmax = 1, nsample - 1
    if cutoff_sigma > 0:
        kmin = max(kmin, int(np.ceil(mean - cutoff_sigma * sd)))
        kmax = min(kmax, int(np.floor(mean + cutoff_sigma * sd)))
    tab = np.zeros((nsample + 1, 2), dtype=np.float32)
    k = np.arange(kmin, kmax + 1, dtype=np.float64)
    phi = k / nsample
    u = np.sqrt(2.0) * erfinv(phi)
    sig = thr / u
    g = np.sqrt(2.0 / np.pi) * u * np.exp(-0.5 * u * u)
    tab[kmin:kmax + 1, 0] = (sig * np.sqrt(1.0 - g / phi)).astype(np.float32)
    tab[kmin:kmax + 1, 1] = (sig * np.sqrt(1.0 + g / (1.0 - phi))).astype(np.float32)
    return tab


def unpack_2bit_dynamic(payload: np.ndarray, nsample: int = DLS_NSAMPLE, cutoff_sigma: float = DLS_CUTOFF_SIGMA,
                        threshold: float = DLS_THRESHOLD) -> np.ndarray:
    """u8[nbytes] -> f64[2][2*nbytes] with the levels of each window looked up by its low-state count (see above).
    The sample count must be a multiple of the window (the filterbank's blocks are)."""
    b = payload.astype(np.uint8)
    st = np.empty((2, b.size * 2), dtype=np.uint8)          # offset-binary states, same bit layout as unpack_2bit
    st[0, 0::2] = b & 3
    st[1, 0::2] = (b >> 2) & 3
    st[0, 1::2] = (b >> 4) & 3
    st[1, 1::2] = (b >> 6) & 3
    assert st.shape[1] % nsample == 0, "sample count must be a multiple of the window"
    low = (st == 1) | (st == 2)
    k = low.reshape(2, -1, nsample).sum(axis=2)             # [pol][window]
    tab = dls_table(nsample, cutoff_sigma, threshold).astype(np.float64)
    lev = tab[k]                                            # [pol][window][2]
    mag = np.where(low.reshape(2, -1, nsample), lev[:, :, 0:1], lev[:, :, 1:2]).reshape(2, -1)
    return np.where(st >= 2, mag, -mag)


LEVELS_1BIT = np.array([-1.0, 1.0], dtype=np.float64)


def unpack_1bit(payload: np.ndarray) -> np.ndarray:
    """u8[nbytes] -> f64[2][4*nbytes]: 1-bit, 2-channel VDIF (mode VDIF_8000-1024-16-1, spif2file.sh:58-61):
    bit 2i = pol0 sample i, bit 2i+1 = pol1 sample i (i = 0..3, LSB first); 0 -> -1, 1 -> +1."""
    b = payload.astype(np.uint8)
    out = np.empty((2, b.size * 4), dtype=np.float64)
    for i in range(4):
        out[0, i::4] = LEVELS_1BIT[(b >> (2 * i)) & 1]
        out[1, i::4] = LEVELS_1BIT[(b >> (2 * i + 1)) & 1]
    return out


def unpack(payload: np.ndarray, bits: int, levels=None, dynamic=None) -> np.ndarray:
    """dynamic: None = static table; dict(nsample=, cutoff_sigma=, threshold=) = dynamic level setting (2 bit only)"""
    if dynamic is not None:
        assert bits == 2, "dynamic level setting needs 2-bit input"
        return unpack_2bit_dynamic(payload, **dynamic)
    return unpack_1bit(payload) if bits == 1 else unpack_2bit(payload, levels)


# --------------------------------------------------------------------------------------------
# Filterbank (A5 + A6): -F C:R
# --------------------------------------------------------------------------------------------
def freq_res_for(nchan: int) -> int:
    """leakage factor rule of process_vdif.py:162."""
    return 512 if nchan <= 128 else 2 * nchan


def filterbank_block(x: np.ndarray, nchan: int, freq_res: int) -> np.ndarray:
    """f64[npol][N] -> c128[npol][nchan][freq_res], N = 2*nchan*freq_res.

    Forward real FFT of N samples, spectrum cut into nchan slices of freq_res bins, backward
    complex FFT (unnormalised) per slice.
    """
    npol, n = x.shape
    assert n == 2 * nchan * freq_res
    spec = np.fft.rfft(x, axis=1)[:, : nchan * freq_res]          # drop Nyquist
    spec = spec.reshape(npol, nchan, freq_res)
    return np.fft.ifft(spec, axis=2) * freq_res                   # unnormalised backward


# --------------------------------------------------------------------------------------------
# Coherent dedispersion inside the filterbank (A6 with -D <dm> -F C:D, process_vdif.py:177-180)
#
# DSPSR (absent, unpinned) multiplies every channel's R spectral bins by a phase-only kernel that
# removes the cold-plasma delay RELATIVE TO THE CHANNEL CENTRE, then discards the samples at both
# ends of each block that the cyclic convolution pollutes (overlap-save).  Restated from the
# published algorithm (Hankins & Rickett 1975; DSPSR's dispersion constant 1/2.41e-4 s MHz^2):
#   delay(nu) = DM / (2.41e-4 nu^2)  [s, nu in MHz]
#   phi(fs)   = 2 pi * 1e6 * DM/2.41e-4 * fs^2 / (nu0^2 (nu0 + fs)),  fs = sky offset from centre nu0
#   kernel    = exp(-i phi) for USB (baseband frequency rises with sky frequency), exp(+i phi) for LSB
# Choices that DSPSR makes internally and that are OURS here (documented in DESIGN.md section 3c):
# the guard on the smearing (5 %), R = smallest power of two >= max(rule of :162, 4 x discarded),
# the kernel is 1 on bin (k=0, j=0) (the real-valued DC term of the band).
# --------------------------------------------------------------------------------------------
DM_DISPERSION = 2.41e-4          # DSPSR's dispersion constant; delay = DM / (DM_DISPERSION * nu_MHz^2) seconds
SMEAR_GUARD = 1.05
MAX_COHERENT_FREQ_RES = 8192


def channel_centres_sky(freq_mhz: float, bw_mhz: float, nchan: int) -> np.ndarray:
    """sky frequency of the centre of filterbank channel k (baseband order, before the USB flip)."""
    df = abs(bw_mhz) / nchan
    k = np.arange(nchan, dtype=np.float64)
    if bw_mhz > 0:
        return freq_mhz - abs(bw_mhz) / 2.0 + (k + 0.5) * df
    return freq_mhz + abs(bw_mhz) / 2.0 - (k + 0.5) * df


def smearing_samples(freq_mhz: float, bw_mhz: float, nchan: int, dm: float):
    """(nfilt_pos, nfilt_neg): channel samples polluted at the start / at the end of a block.

    Worst case = the lowest-frequency channel.  Output sample t needs inputs t - t_hi .. t + t_lo
    (components above the centre arrived t_hi early, below it t_lo late)."""
    df = abs(bw_mhz) / nchan
    nu0 = channel_centres_sky(freq_mhz, bw_mhz, nchan).min()
    d = abs(dm) / DM_DISPERSION
    t_lo = d * (1.0 / (nu0 - df / 2.0) ** 2 - 1.0 / nu0 ** 2)
    t_hi = d * (1.0 / nu0 ** 2 - 1.0 / (nu0 + df / 2.0) ** 2)
    rate = df * 1.0e6
    return int(np.ceil(t_hi * rate * SMEAR_GUARD)), int(np.ceil(t_lo * rate * SMEAR_GUARD))


def coherent_geometry(freq_mhz: float, bw_mhz: float, nchan: int, freq_res: int, tscrunch: int, dm: float):
    """(R, nfilt_pos, nfilt_neg, keep): keep = R - nfilt_pos - nfilt_neg is a multiple of tscrunch."""
    pos, neg = smearing_samples(freq_mhz, bw_mhz, nchan, dm)
    r = freq_res or freq_res_for(nchan)
    while r < 4 * (pos + neg) or r < 2 * tscrunch:
        r *= 2
    if r > MAX_COHERENT_FREQ_RES:
        raise ValueError(f"DM {dm} smears {pos + neg} samples of a {abs(bw_mhz) / nchan} MHz channel: "
                         f"freq_res would exceed {MAX_COHERENT_FREQ_RES}")
    neg += (r - pos - neg) % tscrunch
    return r, pos, neg, r - pos - neg


def chirp(freq_mhz: float, bw_mhz: float, nchan: int, freq_res: int, dm: float) -> np.ndarray:
    """c128[nchan][freq_res] dedispersion kernel, bin j of channel k = big-spectrum bin k*R + j."""
    df = abs(bw_mhz) / nchan
    nu0 = channel_centres_sky(freq_mhz, bw_mhz, nchan)[:, None]
    j = np.arange(freq_res, dtype=np.float64)[None, :]
    fb = (j / freq_res - 0.5) * df                       # baseband offset from the channel centre
    fs = fb if bw_mhz > 0 else -fb                       # sky offset
    phi = 2.0 * np.pi * 1.0e6 * (dm / DM_DISPERSION) * fs * fs / (nu0 * nu0 * (nu0 + fs))
    h = np.exp((-1j if bw_mhz > 0 else 1j) * phi)
    h[0, 0] = 1.0
    return h


def filterbank_block_coherent(x: np.ndarray, nchan: int, freq_res: int, kernel: np.ndarray) -> np.ndarray:
    """as filterbank_block with the kernel applied between the forward and the backward transforms"""
    npol, n = x.shape
    assert n == 2 * nchan * freq_res
    spec = np.fft.rfft(x, axis=1)[:, : nchan * freq_res].reshape(npol, nchan, freq_res)
    return np.fft.ifft(spec * kernel[None], axis=2) * freq_res


# --------------------------------------------------------------------------------------------
# Detection (A7): -P0/-P1/-d1/-d3/-d4
# --------------------------------------------------------------------------------------------
def nif_for(pol_mode: int) -> int:
    return 4 if pol_mode in (4, 5) else 1


def detect(y: np.ndarray, pol_mode: int) -> np.ndarray:
    """c128[2][C][R] -> f64[nif][C][R]  (semantics: process_vdif.py:58-64, base2fil.sh:214-217).

    0/1: |p|^2 of that pol; 2: PP+QQ; 3: (PP+QQ)^2; 4: PP, QQ, Re(p0 p1*), Im(p0 p1*).
    """
    pp = y[0].real ** 2 + y[0].imag ** 2
    qq = y[1].real ** 2 + y[1].imag ** 2
    if pol_mode == 0:
        return pp[None]
    if pol_mode == 1:
        return qq[None]
    if pol_mode == 2:
        return (pp + qq)[None]
    if pol_mode == 3:
        return ((pp + qq) ** 2)[None]
    if pol_mode == 4:
        pq = y[0] * np.conj(y[1])
        return np.stack([pp, qq, pq.real, pq.imag])
    if pol_mode == 5:
        # Stokes parameters of circularly polarised feeds ("BASIS Circular", process_vdif.py:131) from the -d4
        # coherency products: I = PP+QQ, Q = 2 Re(PQ*), U = 2 Im(PQ*), V = PP-QQ.  An extension of the build
        # (north_star "IQUV formation"); the reference itself stops at the products (:58-64,175-176).
        pq = y[0] * np.conj(y[1])
        return np.stack([pp + qq, 2.0 * pq.real, 2.0 * pq.imag, pp - qq])
    raise ValueError(f"pol = {pol_mode} not implemented. Choices are 0, 1, 2, 3, 4")


# --------------------------------------------------------------------------------------------
# Time scrunch (A8): -t T
# --------------------------------------------------------------------------------------------
def tscrunch(p: np.ndarray, factor: int) -> np.ndarray:
    """f64[nif][C][nt] -> f64[nif][C][nt // factor]  (sum of `factor` adjacent samples)."""
    if factor <= 1:
        return p
    nif, c, nt = p.shape
    nt2 = nt // factor
    return p[:, :, : nt2 * factor].reshape(nif, c, nt2, factor).sum(axis=3)


# --------------------------------------------------------------------------------------------
# Rescale (A9): -c / -I
# --------------------------------------------------------------------------------------------
def rescale_stats(p: np.ndarray):
    """f64[nif][C][nt] -> (offset[nif][C], scale[nif][C]); offset=-mean, scale=1/sigma."""
    mean = p.mean(axis=2)
    var = (p * p).mean(axis=2) - mean * mean
    scale = np.where(var > 0.0, 1.0 / np.sqrt(np.where(var > 0.0, var, 1.0)), 1.0)
    return -mean, scale


def rescale_apply(p: np.ndarray, offset: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """out = (in + offset) * scale, evaluated in fp32 on fp32 data with fp32 offset/scale (the
    stage works on float time series; only the statistics are accumulated in double)."""
    p32 = p.astype(np.float32)
    return (p32 + offset.astype(np.float32)[:, :, None]) * scale.astype(np.float32)[:, :, None]


# --------------------------------------------------------------------------------------------
# Digitise (A10): -b {2,8,16,-32}
# --------------------------------------------------------------------------------------------
def digi_params(nbit: int):
    if nbit == 2:
        return 1.5, 1.0, 3
    if nbit == 8:
        return 127.5, 127.5 / DIGI_SIGMA, 255
    if nbit == 16:
        return 32767.5, 32767.5 / DIGI_SIGMA, 65535
    if nbit == -32:
        return 0.0, 1.0, None
    raise ValueError(f"nbit={nbit} not in supported values of [2, 8, 16, -32]. ")


def digitise_values(x: np.ndarray, nbit: int) -> np.ndarray:
    """Rescaled floats -> integer codes (before bit packing); float32 for -32."""
    mean, scale, vmax = digi_params(nbit)
    if nbit == -32:
        return x.astype(np.float32)
    # the C expression `int(x * digi_scale + digi_mean + 0.5)` with float x, digi_scale, digi_mean:
    # product and first sum are fp32, the 0.5 literal promotes the last sum to double (exact)
    t = x.astype(np.float32) * np.float32(scale) + np.float32(mean)
    v = np.clip(t.astype(np.float64) + 0.5, 0.0, float(vmax))   # clip first: no integer overflow
    v = np.trunc(v)
    return v.astype(np.uint16 if nbit == 16 else np.uint8)


def pack_codes(codes: np.ndarray, nbit: int) -> bytes:
    """codes in output order (flattened [t][nif][chan]) -> byte stream. 2-bit: 4 per byte LSB first."""
    flat = codes.reshape(-1)
    if nbit == 2:
        assert flat.size % 4 == 0
        q = flat.reshape(-1, 4).astype(np.uint8)
        return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8).tobytes()
    if nbit == 8:
        return flat.astype(np.uint8).tobytes()
    if nbit == 16:
        return flat.astype("<u2").tobytes()
    return flat.astype("<f4").tobytes()


# --------------------------------------------------------------------------------------------
# SIGPROC header (A10, public sigproc format)
# --------------------------------------------------------------------------------------------
SIGPROC_TELESCOPE_ID = {
    "fake": 0, "arecibo": 1, "ooty": 2, "nancay": 3, "parkes": 4, "jodrell": 5, "gbt": 6,
    "gmrt": 7, "effelsberg": 8, "ata": 9, "srt": 10, "lofar": 11, "vla": 12,
}


def _sp_str(s: str) -> bytes:
    b = s.encode("ascii")
    return struct.pack("<i", len(b)) + b


def sigproc_ra_dec(text: str) -> float:
    """'hh:mm:ss.ss' / 'dd:mm:ss.ss' -> sigproc packed double (hhmmss.ss)."""
    t = text.strip()
    sign = -1.0 if t.startswith("-") else 1.0
    t = t.lstrip("+-")
    parts = t.split(":")
    parts += ["0"] * (3 - len(parts))
    return sign * (float(parts[0]) * 10000.0 + float(parts[1]) * 100.0 + float(parts[2]))


def sigproc_header(*, telescope: str, source: str, ra: str, dec: str, rawdatafile: str,
                   tstart_mjd: float, tsamp_s: float, nbits: int, fch1: float, foff: float,
                   nchans: int, nifs: int, refdm: float = 0.0) -> bytes:
    h = _sp_str("HEADER_START")
    h += _sp_str("telescope_id") + struct.pack("<i", SIGPROC_TELESCOPE_ID.get(telescope.lower(), 0))
    h += _sp_str("machine_id") + struct.pack("<i", 0)
    h += _sp_str("data_type") + struct.pack("<i", 1)
    h += _sp_str("rawdatafile") + _sp_str(rawdatafile[-80:])
    h += _sp_str("source_name") + _sp_str(source[:80])
    h += _sp_str("barycentric") + struct.pack("<i", 0)
    h += _sp_str("pulsarcentric") + struct.pack("<i", 0)
    h += _sp_str("az_start") + struct.pack("<d", 0.0)
    h += _sp_str("za_start") + struct.pack("<d", 0.0)
    h += _sp_str("src_raj") + struct.pack("<d", sigproc_ra_dec(ra))
    h += _sp_str("src_dej") + struct.pack("<d", sigproc_ra_dec(dec))
    h += _sp_str("tstart") + struct.pack("<d", tstart_mjd)
    h += _sp_str("tsamp") + struct.pack("<d", tsamp_s)
    h += _sp_str("nbits") + struct.pack("<i", 32 if nbits == -32 else nbits)
    h += _sp_str("fch1") + struct.pack("<d", fch1)
    h += _sp_str("foff") + struct.pack("<d", foff)
    h += _sp_str("nchans") + struct.pack("<i", nchans)
    h += _sp_str("nifs") + struct.pack("<i", nifs)
    h += _sp_str("refdm") + struct.pack("<d", refdm)
    h += _sp_str("HEADER_END")
    return h


def channel_freqs(freq_mhz: float, bw_mhz: float, nchan: int):
    """(fch1, foff) of the OUTPUT order (descending sky frequency)."""
    df = abs(bw_mhz) / nchan
    top_edge = freq_mhz + abs(bw_mhz) / 2.0
    return top_edge - df / 2.0, -df


# --------------------------------------------------------------------------------------------
# Whole path
# --------------------------------------------------------------------------------------------
@dataclass
class Config:
    freq_mhz: float = 1608.0
    bw_mhz: float = 16.0            # signed: negative = LSB (process_vdif.py:117-118)
    start_s: float = 0.0            # -S
    total_s: float = 10.0           # -T
    nchan: int = 128                # -F C:
    freq_res: int = 0               # -F :R   (0 -> rule of process_vdif.py:162)
    tscrunch: int = 1               # -t
    nbit: int = 8                   # -b
    pol_mode: int = 2               # -P0/-P1 -> 0/1 ; -d1 -> 2 ; -d3 -> 3 ; -d4 -> 4
    rescale_constant: bool = True   # -c
    rescale_interval_s: float = DEFAULT_RESCALE_INTERVAL_S  # -I ; 0 = disabled
    dm: float = 0.0
    coherent: bool = False          # -F C:D (process_vdif.py:177-180): dedisperse inside the filterbank
    telescope: str = "ONSALA85"
    source: str = "J0000+0000"
    ra: str = "00:00:00.0"
    dec: str = "00:00:00.0"
    rawdatafile: str = ""
    levels: tuple | None = None     # 2-bit level table (None = DSPSR's static one)
    dynamic: dict | None = None     # dynamic level setting instead: dict(nsample=, cutoff_sigma=, threshold=) (unpack_2bit_dynamic)
    fixed_offset: np.ndarray | None = None   # set_rescale equivalent
    fixed_scale: np.ndarray | None = None
    result: dict = field(default_factory=dict)


def detected_power(raw_frames: np.ndarray, cfg: Config):
    """Frames (uint8, starting at a frame boundary) -> f64[nif][C][nt] after tscrunch, plus header.

    Applies -S / -T (whole seconds counted in samples), drops the partial last block.
    """
    hdr = parse_vdif_header(raw_frames[:32].tobytes())
    rate = 2.0e6 * abs(cfg.bw_mhz)                    # real samples / s / pol
    payload, bad_frames, counters = assemble_stream(raw_frames, rate)
    cfg.result["frame_counters"] = counters
    c = cfg.nchan
    r = cfg.freq_res or freq_res_for(c)
    pos = neg = 0
    keep = r
    if cfg.coherent:
        r, pos, neg, keep = coherent_geometry(cfg.freq_mhz, cfg.bw_mhz, c, cfg.freq_res, cfg.tscrunch, cfg.dm)
        kernel = chirp(cfg.freq_mhz, cfg.bw_mhz, c, r, cfg.dm)
    n = 2 * c * r
    hop = 2 * c * keep                                # overlap-save: blocks advance by the kept samples
    bits = hdr.bits_per_sample
    spb = 4 // bits                                   # dual-pol time samples per payload byte
    s0 = int(round(cfg.start_s * rate))
    s0 -= s0 % spb                                    # byte aligned
    navail = payload.size * spb - s0
    nwant = int(round(cfg.total_s * rate))
    nsamp = max(0, min(navail, nwant))
    nblocks = (nsamp - n) // hop + 1 if nsamp >= n else 0
    b_lo = s0 // spb
    b_hi = b_lo + ((nblocks - 1) * hop + n) // spb if nblocks else b_lo
    x = unpack(payload[b_lo:b_hi], bits, cfg.levels, cfg.dynamic)
    if bad_frames.any() and b_hi > b_lo:              # invalid / filler frames: zero voltages
        pbytes = hdr.payload_bytes
        byte_bad = np.repeat(bad_frames, pbytes)[b_lo:b_hi]
        x[:, np.repeat(byte_bad, spb)] = 0.0
    out = []
    for b in range(nblocks):
        xb = x[:, b * hop: b * hop + n]
        if cfg.coherent:
            y = filterbank_block_coherent(xb, c, r, kernel)[:, :, pos: r - neg]
        else:
            y = filterbank_block(xb, c, r)
        out.append(detect(y, cfg.pol_mode))
    nif = nif_for(cfg.pol_mode)
    p = np.concatenate(out, axis=2) if out else np.zeros((nif, c, 0))
    cfg.result["geometry"] = (r, pos, neg, keep)
    return tscrunch(p, cfg.tscrunch), hdr, s0 + 2 * c * pos, r


def channelise(raw_frames: np.ndarray, cfg: Config) -> bytes:
    """Whole path: frames -> SIGPROC .fil bytes (header + samples)."""
    p, hdr, s0, r = detected_power(raw_frames, cfg)
    nif, c, nt = p.shape
    tsamp_s = c * cfg.tscrunch / (abs(cfg.bw_mhz) * 1.0e6)
    rate = 2.0e6 * abs(cfg.bw_mhz)
    fps = rate * 2 * hdr.bits_per_sample / 8 / hdr.payload_bytes   # frames per second for 2 pol x nbit
    tstart = (vdif_epoch_mjd(hdr.ref_epoch)
              + (hdr.seconds + hdr.frame_nr / fps + s0 / rate) / 86400.0)

    offset = np.zeros((nif, c))
    scale = np.ones((nif, c))
    if cfg.fixed_offset is not None:
        offset, scale = cfg.fixed_offset.astype(np.float64), cfg.fixed_scale.astype(np.float64)
        x = rescale_apply(p, offset, scale)
    elif cfg.rescale_interval_s > 0 and nt > 0:
        rate_out = abs(cfg.bw_mhz) * 1.0e6 / (c * cfg.tscrunch)     # output samples per second
        nint = max(1, int(cfg.rescale_interval_s * rate_out))       # truncation, as a uint64 cast
        x = np.empty_like(p)
        for i0 in range(0, nt, nint):
            seg = p[:, :, i0:i0 + nint]
            if i0 == 0 or not cfg.rescale_constant:
                offset, scale = rescale_stats(seg)
            x[:, :, i0:i0 + nint] = rescale_apply(seg, offset, scale)
            if i0 == 0:
                cfg.result["offset0"], cfg.result["scale0"] = offset.copy(), scale.copy()
    else:
        x = p
    cfg.result["power"] = p
    cfg.result["rescaled"] = x

    if cfg.bw_mhz > 0:                                # USB: flip so that foff < 0
        x = x[:, ::-1, :]
    codes = digitise_values(np.ascontiguousarray(x.transpose(2, 0, 1)), cfg.nbit)  # [t][nif][chan]
    cfg.result["codes"] = codes
    fch1, foff = channel_freqs(cfg.freq_mhz, cfg.bw_mhz, c)
    head = sigproc_header(telescope=cfg.telescope, source=cfg.source, ra=cfg.ra, dec=cfg.dec,
                          rawdatafile=cfg.rawdatafile, tstart_mjd=tstart, tsamp_s=tsamp_s,
                          nbits=cfg.nbit, fch1=fch1, foff=foff, nchans=c, nifs=nif, refdm=cfg.dm)
    return head + pack_codes(codes, cfg.nbit)


def config_from_hdr(hdr_path: str, **kw) -> Config:
    h = read_hdr(hdr_path)
    return Config(freq_mhz=h["FREQ"], bw_mhz=h["BW"], telescope=h["TELESCOPE"],
                  source=h["SOURCE"], ra=h["RA"], dec=h["DEC"], rawdatafile=h["DATAFILE"], **kw)
