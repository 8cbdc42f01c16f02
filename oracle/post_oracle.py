"""CPU oracle (numpy) for the stages behind the filterbank.  TEST INFRASTRUCTURE ONLY (see frb_oracle.py).

PARITY UNPINNED: the reference runs PRESTO's ``prepdata`` / ``prepsubband`` (process_vdif.py:202-229) and DSPSR's
``dspsr -E <par> -L 10 -A -d1 <IFall.fil>`` (base2fil.sh:474) here; neither program is in /root/reference or in this
image, and the reference holds no outputs of them.  The functions below restate the published algorithms with the
conventions include/frbch.h documents:

  stage                 reference anchor                          function
  --------------------  ----------------------------------------  ------------------
  DM list               process_vdif.py:209-214                   dm_list (in the product's post.py: pure bookkeeping, golden-tested)
  delays                prepdata -dm / -lodm -numdms -dmstep      delays_samples: DM / 2.41e-4 (f^-2 - f_top^-2) / tsamp, int(x + 0.5)
  -clip <sigma>         process_vdif.py:223-224                   clip_flags: two rounds of mean / sigma on the zero-DM sum
  -zerodm               process_vdif.py:221-222                   subtract the mean over channels of every time sample
  dedispersed series    -o <outfile>                              dedisperse
  fold                  base2fil.sh:474                           fold: turns = F0 tau + F1 tau^2 / 2 about PEPOCH, topocentric
"""
from __future__ import annotations

import numpy as np

DM_CONST = 1.0 / 2.41e-4
CHUNK_ROWS = 4096


def chan_freqs(fch1, foff, nchan):
    return fch1 + np.arange(nchan, dtype=np.float64) * foff


def delays_seconds(fch1, foff, nchan, dm):
    fc = chan_freqs(fch1, foff, nchan)
    fhi = fch1 if foff < 0 else fc[-1]
    return dm * DM_CONST * (1.0 / (fc * fc) - 1.0 / (fhi * fhi))


def delays_samples(fch1, foff, nchan, tsamp, dm):
    return np.floor(delays_seconds(fch1, foff, nchan, dm) / tsamp + 0.5).astype(np.int64)


def seq_sum(x, axis):
    """sum along an axis in index order (np.add.accumulate is sequential), as the kernels do"""
    return np.take(np.add.accumulate(x, axis=axis), -1, axis=axis)


def clip_flags(S, clip_sigma):
    flag = np.zeros(S.size, dtype=bool)
    for _ in range(2):
        good = S[~flag]
        if good.size == 0:
            break
        s1 = seq_sum(good, 0)
        s2 = seq_sum(good * good, 0)
        mean = s1 / good.size
        var = s2 / good.size - mean * mean
        sig = np.sqrt(var) if var > 0 else 0.0
        flag = np.abs(S - mean) > clip_sigma * sig
    return flag


def dedisperse(x, *, fch1, foff, tsamp, dms, zerodm=True, clip=5.0, integer=True):
    """x: [nrows][nchan] samples (one product).  -> (float32 [ndm][nout], number of clipped rows)"""
    x = np.asarray(x, dtype=np.float64)
    nrows, nchan = x.shape
    S = seq_sum(x, 1)
    nclip = 0
    if clip > 0:
        flag = clip_flags(S, clip)
        nclip = int(flag.sum())
        if 0 < nclip < nrows:
            xg = np.where(flag[:, None], 0.0, x)
            part = [seq_sum(xg[r0:r0 + CHUNK_ROWS], 0) for r0 in range(0, nrows, CHUNK_ROWS)]
            m = seq_sum(np.stack(part), 0)
            ngood = float(nrows - nclip)
            repl = np.floor(m / ngood + 0.5) if integer else m / ngood
            x = np.where(flag[:, None], repl[None, :], x)
            S = seq_sum(x, 1)
        else:
            nclip = 0
    dl = [delays_samples(fch1, foff, nchan, tsamp, dm) for dm in dms]
    nout = nrows - max(int(d.max()) for d in dl)
    out = np.empty((len(dms), nout), dtype=np.float32)
    for i, d in enumerate(dl):
        a = np.zeros(nout)
        b = np.zeros(nout)
        for c in range(nchan):                       # ascending channel order, as the kernel
            a += x[d[c]: d[c] + nout, c]
            if zerodm:
                b += S[d[c]: d[c] + nout]
        out[i] = (a - b / nchan if zerodm else a).astype(np.float32)
    return out, nclip


def fold(x, *, fch1, foff, tsamp, tstart_mjd, f0, f1, pepoch_mjd, dm, nbin, subint_s, apply_delays):
    """x: [nrows][nchan].  -> (sums float64 [nsub][nchan][nbin], hits uint32)"""
    x = np.asarray(x, dtype=np.float64)
    nrows, nchan = x.shape
    rps = max(1, int(round(subint_s / tsamp)))
    nsub = (nrows + rps - 1) // rps
    t = np.arange(nrows, dtype=np.float64)
    tau0 = (tstart_mjd - pepoch_mjd) * 86400.0 + t * tsamp
    dly = delays_seconds(fch1, foff, nchan, dm) if (apply_delays and dm != 0.0) else None
    sub = (np.arange(nrows) // rps).astype(np.int64)
    prof = np.zeros((nsub, nchan, nbin))
    hits = np.zeros((nsub, nchan, nbin), dtype=np.uint32)
    half_f1 = 0.5 * f1
    for c in range(nchan):
        tau = tau0 - dly[c] if dly is not None else tau0
        turns = f0 * tau + (half_f1 * tau) * tau
        fr = turns - np.floor(turns)
        b = np.minimum((fr * nbin).astype(np.int64), nbin - 1)
        flat = sub * nbin + b
        prof[:, c, :] = np.bincount(flat, weights=x[:, c], minlength=nsub * nbin).reshape(nsub, nbin)
        hits[:, c, :] = np.bincount(flat, minlength=nsub * nbin).reshape(nsub, nbin)
    return prof, hits


# --------------------------------------------------------------------------------------------------------------------
# corner turn (spif2file.sh:31-113): jive5ab is absent; the bit order restated here is the contract of frbch_cornerturn_*
# --------------------------------------------------------------------------------------------------------------------
def parse_recipe(recipe: str):
    swap = recipe.startswith("swap_sign_mag+")
    if swap:
        recipe = recipe[len("swap_sign_mag+"):]
    w, rest = recipe.split(">", 1)
    body, _, tags = rest.partition(":")
    groups = [[int(b) for b in g.split(",")] for g in body.strip("[]").split("][")]
    return int(w), groups, swap


def cornerturn(payload: np.ndarray, recipe: str):
    """payload bytes of the recorder stream -> list of per-tag payload byte arrays (output bit k of group g of word i =
    input bit groups[g][k] of word i; everything LSB first)"""
    w, groups, swap = parse_recipe(recipe)
    bits = np.unpackbits(payload, bitorder="little")
    words = bits.reshape(-1, w)
    if swap:
        words = words.reshape(-1, w // 2, 2)[:, :, ::-1].reshape(-1, w)
    return [np.packbits(words[:, g].reshape(-1), bitorder="little") for g in groups]


def interleave(streams, recipe: str):
    """inverse of cornerturn: per-tag payloads -> recorder payload (unlisted bits zero)"""
    w, groups, swap = parse_recipe(recipe)
    glen = len(groups[0])
    nwords = streams[0].size * 8 // glen
    words = np.zeros((nwords, w), np.uint8)
    for g, s in zip(groups, streams):
        words[:, g] = np.unpackbits(s, bitorder="little").reshape(nwords, glen)
    if swap:
        words = words.reshape(-1, w // 2, 2)[:, :, ::-1].reshape(-1, w)
    return np.packbits(words.reshape(-1), bitorder="little")
