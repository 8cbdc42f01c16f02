import sys, numpy as np
sys.path.insert(0, '.')
from frb_baseband_amd import channeliser as ch, synth, sigproc, _lib
from oracle import frb_oracle as o
from tests import parity_util as pu
from tests.hipmem import DeviceBuffer
lib = _lib.load()
bw, nchan, secs, pol, nbit, tscr, flags = 32.0, 1024, 0.27, 2, 8, 1, 0
raw = synth.make_vdif(secs, bw_mhz=abs(bw), nchan=nchan)
d_raw = DeviceBuffer.from_numpy(raw)
nfr = raw.size // 8032
cfg = pu.lib_cfg(lib, bw, nchan, secs, pol=pol, nbit=nbit, tscr=tscr, flags=flags)
with ch.Channeliser(cfg, lib) as c:
    info = c.info
    nblocks = (nfr * 8000) // info.block_payload_bytes
    rows = nblocks * info.rows_per_block
    pw = DeviceBuffer(rows * nchan * 4)
    c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
    power = pw.to_numpy(np.float32).reshape(rows, 1, nchan)
    c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
    power2 = pw.to_numpy(np.float32).reshape(rows, 1, nchan)
    print("power repeatable:", np.array_equal(power, power2))
    out = DeviceBuffer(rows * info.row_bytes)
    r1 = c.process_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, out.ptr.value, out.nbytes)
    r2 = c.flush_device(out.ptr.value + r1 * info.row_bytes, out.nbytes - r1 * info.row_bytes)
    buffered = out.to_numpy(np.uint8).reshape(rows, 1, nchan)
    off, sc = c.get_rescale()
offo, sco = off[:, ::-1], sc[:, ::-1]
x = (power.transpose(1, 2, 0) + offo.astype(np.float32)[:, :, None]) * sco.astype(np.float32)[:, :, None]
x = np.ascontiguousarray(x.transpose(2, 0, 1))
want = o.digitise_values(x, 8)
bad = np.argwhere(buffered != want)
print("nbad", len(bad))
for b in bad[:10]:
    t_, p_, k_ = b
    pv_ = power[t_, p_, k_]; xx = x[t_, p_, k_]
    tt = np.float32(xx) * np.float32(127.5 / 6) + np.float32(127.5)
    print(b, "power %.9g off %.9g sc %.9g x %.9g t %.9g got %d want %d" % (pv_, offo[p_, k_], sco[p_, k_], xx, tt, buffered[t_, p_, k_], want[t_, p_, k_]))
