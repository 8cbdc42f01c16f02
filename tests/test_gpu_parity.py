"""GPU parity tests proper: the HIP path, called through the C ABI, against the fp64 oracle."""
import numpy as np
import pytest

from tests import parity_util as pu

pytestmark = pytest.mark.gpu

CASES = [
    # bw, nchan, secs, kwargs
    (16.0, 128, 0.05, {}),                                   # BASELINE config 1 shape (-F128:512 -d1)
    (-16.0, 128, 0.05, {}),                                  # LSB: no band flip
    (16.0, 128, 0.05, dict(flags=2)),                        # the same shape with the generic K2 behind the fast K1
    (16.0, 128, 0.05, dict(pol=4, tscr=8)),                  # 2C = 256 K2 (radix 16 x 16, four sequences per wave), 2 waves per workgroup
    (-16.0, 128, 0.05, dict(tscr=16, nbit=16)),              # ... 4 waves
    (16.0, 128, 0.05, dict(pol=4, tscr=32, nbit=-32)),       # ... 8 waves
    (16.0, 128, 0.05, dict(pol=0, nbit=2, tscr=2)),
    (16.0, 128, 0.05, dict(pol=3, interval=0.01, const=0)),  # per-interval rescale (pol 3: separate statistics pass)
    (16.0, 64, 0.05, dict(pol=4)),                           # -d4 coherency products
    (-16.0, 32, 0.05, dict(pol=4, nbit=-32, tscr=4, freq_res=64)),
    (16.0, 32, 0.05, dict(pol=0, nbit=2, tscr=2, freq_res=64)),
    (16.0, 32, 0.05, dict(pol=3, nbit=16, freq_res=64, interval=0.0)),   # -I0 (keepBP)
    (16.0, 32, 0.05, dict(pol=1, freq_res=64, interval=0.004, const=0)),  # no -c: per-interval rescale
    (16.0, 32, 0.05, dict(freq_res=64, interval=0.004, const=1, maxb=3)),
    (32.0, 1024, 0.14, {}),                                  # BASELINE config 2 shape, 2 blocks (fast K1+K2, M=8)
    (32.0, 1024, 0.27, dict(maxb=3)),
    (32.0, 1024, 0.14, dict(flags=3)),                       # same through the generic kernels
    (32.0, 1024, 0.14, dict(flags=1)),                       # generic K1 + fast K2
    (32.0, 1024, 0.14, dict(flags=2)),                       # fast K1 + generic K2
    (-32.0, 1024, 0.14, dict(pol=4, tscr=2)),                # BASELINE config 3 shape (-d4), LSB, -t 2
    # two-pass rescale of a first interval (flag 1 << 28; automatic with four products): frbch_k2_priv sums only, then digitises the same resident spill
    (32.0, 1024, 0.14, dict(flags=1 << 28)),                 # the scan ends inside its first interval: second pass at the flush
    (-32.0, 1024, 0.27, dict(flags=1 << 28, pol=5, interval=0.1)),       # the interval ends inside the (only) batch: statistics over its rows, codes for all
    (32.0, 1024, 0.27, dict(flags=1 << 28, pol=4, tscr=2, nbit=16, maxb=3)),   # a second batch arrives while the first is deferred: its float rows are written after all
    (32.0, 1024, 0.27, dict(flags=1 << 28, pol=2, tscr=4, nbit=2, interval=0.1, maxb=2)),   # interval end inside the second batch: buffered form
    (-32.0, 1024, 0.14, dict(pol=4, tscr=4, nbit=-32)),
    (32.0, 1024, 0.14, dict(pol=0, nbit=2, tscr=4)),
    (32.0, 1024, 0.14, dict(tscr=8)),                        # -t 8: 8-sequence K2 workgroups
    (-32.0, 1024, 0.14, dict(pol=4, tscr=8, nbit=16)),
    (32.0, 1024, 0.14, dict(tscr=16)),                       # -t 16: two-stage tscrunch (wave K2 with its 8-sample tile + frbch_k2_scrunch x2)
    (-32.0, 1024, 0.28, dict(tscr=32, pol=4, nbit=16)),      # the same x4 with four products
    (64.0, 2048, 0.3, dict(tscr=8)),                         # 2C = 4096: 4-sample tile x2
    (16.0, 128, 0.1, dict(tscr=64, nbit=2)),                 # 2C = 256 (four sequences per wave): 32-sample tile x2
    (32.0, 1024, 0.14, dict(tscr=16, flags=2)),              # -t 16 on the generic K2
    (32.0, 1024, 0.4, dict(tscr=16, maxb=1, interval=0.2)),  # two-stage tscrunch, one block per launch, the rescale interval ends inside the scan (float rows, then codes)
    (16.0, 512, 0.08, dict(tscr=8, pol=4)),
    (32.0, 1024, 0.14, dict(pol=1, nbit=16, interval=0.0)),
    (32.0, 1024, 0.14, dict(pol=3, interval=0.05, const=0)),
    (32.0, 1024, 0.2, dict(start=0.05, maxb=2)),             # -S, odd batching
    (16.0, 256, 0.04, {}),                                   # M = 2  (R = 2C = 512)
    (16.0, 256, 0.04, dict(pol=4, tscr=8)),
    (-16.0, 512, 0.08, dict(tscr=2)),                        # M = 4  (R = 2C = 1024)
    (64.0, 2048, 0.3, dict(tscr=2)),                         # M = 16 (R = 2C = 4096), config-5 channel count
    (64.0, 2048, 0.3, dict(pol=4, tscr=4)),                  # tscrunch > fast K2 tile: generic K2 + fast K1
    (32.0, 512, 0.1, dict(freq_res=2048)),                   # R != 2C: fast K1 (M=8) + fast K2 (M=4)
    (32.0, 1024, 0.2, dict(start=10 / 64e6)),                # -S not on a 4-byte boundary: generic K1 feeds the fast K2
    (32.0, 1024, 0.2, dict(start=2 / 64e6, pol=4, tscr=2)),
    (32.0, 1024, 0.2, dict(bits=1, pol=4, tscr=2)),          # 1-bit mode VDIF_8000-1024-16-1: generic K1 gather feeds the fast K2
    (-16.0, 128, 0.05, dict(bits=1, start=0.0101)),
    (32.0, 1024, 0.2, dict(payload_bytes=10000)),            # Mark5B-sized payload through the fast kernels
    # coherent dedispersion (-D <dm> -F C:D, process_vdif.py:177-180): K1 forward -> K2c chirp -> K3 -> K4
    (16.0, 16, 0.012, dict(dm=1.0, coherent=1, freq=316.0)),
    (-32.0, 1024, 0.3, dict(dm=56.7, coherent=1, freq=400.0, tscr=4)),
    (-32.0, 512, 0.15, dict(dm=26.7, coherent=1, freq=350.0, pol=4, tscr=2, nbit=16)),
    (32.0, 2048, 0.6, dict(dm=56.7, coherent=1, freq=1400.0, freq_res=4096)),   # BASELINE config 5 shape: -F2048:4096 -D 56.7 -F2048:D (register-pass K1 / K2c / K3, M = 16)
    (32.0, 2048, 0.6, dict(dm=56.7, coherent=1, freq=1400.0, freq_res=4096, flags=3)),   # the same on the generic kernels
    (-32.0, 2048, 0.6, dict(dm=56.7, coherent=1, freq=1400.0, freq_res=4096, pol=5, tscr=2, nbit=16)),   # wave K1 / K3 at R = 4096: four products (IQUV), -t 2, LSB, sums fused in K3
    (32.0, 2048, 0.6, dict(dm=30.0, coherent=1, freq=1400.0, freq_res=4096, pol=0, tscr=4, nbit=2, interval=0.2)),   # ... one product, -t 4, 2-bit codes, the rescale interval ends inside the scan
    (-32.0, 1024, 0.3, dict(dm=56.7, coherent=1, freq=400.0, tscr=4, flags=1)),  # generic K1 / K3 (bit-reversed bins) + register-pass K2c
    (-32.0, 1024, 0.3, dict(dm=56.7, coherent=1, freq=400.0, pol=4, tscr=2, flags=2)),  # register-pass K1 / K3 + generic K2c
    (32.0, 2048, 0.6, dict(dm=56.7, coherent=1, freq=1400.0, start=2 / 64e6)),  # -S off the K1 piece boundary: falls back to the generic K1 / K3, kernel table rebuilt
    (16.0, 256, 0.2, dict(dm=20.0, coherent=1, freq=600.0, pol=0, nbit=16)),     # M = 2 / 2
    (64.0, 4096, 1.1, dict(tscr=8)),                         # BASELINE config 4 shape (-t 8 -F4096:8192), 2 blocks: M = 32 wave kernels, two-stage tscrunch (K2 rows of two samples + frbch_k2_scrunch)
    (-64.0, 4096, 0.55, dict(tscr=4, nbit=2)),                # two-stage tscrunch, factor 2, 2-bit codes, LSB
    (64.0, 4096, 0.55, dict(tscr=8, flags=3)),                # the same through the generic kernels
    (-64.0, 4096, 0.55, {}),                                  # M = 32, -t 1, LSB: wave K2 (frbch_k2_wave<5,8,2,4>, two time samples per workgroup, sums fused)
    (64.0, 4096, 0.55, dict(tscr=2, nbit=-32)),               # the same with -t 2 (one output row per tile)
    (64.0, 4096, 0.55, dict(pol=1, nbit=16)),                 # the same kernel family, single-product instantiation, USB
    (64.0, 4096, 0.55, dict(flags=1 << 20, nbit=2)),          # wave K2 with the separate statistics pass
    (64.0, 4096, 0.55, dict(pol=4, tscr=2, nbit=16)),         # M = 32, coherency products
    (64.0, 4096, 0.55, dict(pol=4, tscr=4)),                  # four products and tscrunch > 2: barrier K2 (two-sample rows) + frbch_k2_scrunch (round 3; generic K2 before)
    (-64.0, 4096, 0.55, dict(pol=5, tscr=8, nbit=16)),        # ... the IQUV spelling of config 4's `-t 8`
    (64.0, 4096, 1.1, dict(pol=4, tscr=8, interval=0.6, maxb=1)),   # ... the rescale interval ends inside the scan, one block per launch (2 blocks)
    # Stokes I,Q,U,V (pol_mode 5, the `-d4 -iquv` extension; north_star "IQUV formation") through every K2 family
    (32.0, 1024, 0.14, dict(pol=5)),                         # wave K2, MSTAT instantiation while the interval is measured
    (-32.0, 1024, 0.14, dict(pol=5, tscr=2, nbit=16)),
    (32.0, 1024, 0.14, dict(pol=5, flags=2)),                # generic K2
    (16.0, 128, 0.05, dict(pol=5, tscr=8, nbit=-32)),        # 2C = 256 wave K2
    (64.0, 4096, 0.55, dict(pol=5, tscr=2)),                  # M = 32
    (-32.0, 512, 0.15, dict(dm=26.7, coherent=1, freq=350.0, pol=5, tscr=2)),   # K3 (register passes)
    # the online chain's upper limit, 2^13 channels (submit_job.py:42,61-71): 2C = 16384 runs the generic radix-2 K2 (one sequence fills
    # the LDS); K1 is the wave kernel up to freq_res 8192 and generic at the default freq_res = 2 C = 16384 (DESIGN.md sections 8.1, 9)
    (64.0, 8192, 0.15, dict(freq_res=512)),                   # wave K1 (M = 2) + generic K2, 2 blocks of 2^23 samples
    (-64.0, 8192, 0.15, dict(freq_res=1024, pol=4, nbit=16)),   # (four products AND -t > 1 at 8192 channels: InputError, the generic K2's accumulators do not fit the LDS)
    # few channels per IF (the online chain's 32 / 64, submit_job.py:74-105; R = 512 as process_vdif.py:162 says): wave K1 (M = 2) +
    # frbch_k2_lane (a whole 64-point across-branch sequence per lane; 2C = 128: per lane pair)
    (16.0, 32, 0.02, {}),
    (-16.0, 32, 0.02, dict(pol=5, tscr=4, nbit=16)),
    (32.0, 32, 0.02, dict(pol=4, nbit=2)),
    (16.0, 32, 0.02, dict(pol=3, nbit=-32, tscr=64)),
    (16.0, 32, 0.02, dict(pol=0, interval=0.004, const=0, maxb=5)),
    (32.0, 64, 0.02, {}),
    (-32.0, 64, 0.02, dict(pol=4, tscr=2)),
    (16.0, 64, 0.02, dict(pol=5, nbit=-32, tscr=32)),
    (16.0, 64, 0.02, dict(pol=1, nbit=2, tscr=8, interval=0.005)),
    (16.0, 64, 0.02, dict(tscr=64)),                          # beyond the lane pair's 32 samples: generic K2
]


@pytest.mark.parametrize("bw,nchan,secs,kw", CASES)
def test_fil_matches_oracle(hip_lib, bw, nchan, secs, kw):
    pu.run_streaming_case(hip_lib, bw, nchan, secs, **kw)


def test_power_tap_matches_oracle(hip_lib):
    """float32 power of the fused unpack->FFT->detect stream vs the fp64 oracle (tolerance stated in parity_util)."""
    from frb_baseband_amd import channeliser as ch, synth
    from tests.hipmem import DeviceBuffer
    from oracle import frb_oracle as o
    bw, nchan = 32.0, 1024
    raw = synth.make_vdif(0.14, bw_mhz=bw, nchan=nchan)
    ocfg = pu.oracle_cfg(bw, nchan, 0.14, pol=4)
    o.channelise(raw, ocfg)
    want = ocfg.result["power"][:, ::-1, :].transpose(2, 0, 1)          # [t][prod][chan], USB flipped
    d_raw = DeviceBuffer.from_numpy(raw)
    with ch.Channeliser(pu.lib_cfg(hip_lib, bw, nchan, 0.14, pol=4), hip_lib) as c:
        info = c.info
        nfr = raw.size // 8032
        nblocks = (nfr * 8000) // info.block_payload_bytes
        pw = DeviceBuffer(nblocks * info.rows_per_block * 4 * nchan * 4)
        c.power_device(d_raw.ptr.value, nfr, 8032, 32, 0, nblocks, pw.ptr.value, pw.nbytes)
        got = pw.to_numpy(np.float32).reshape(want.shape).astype(np.float64)
    scale = (ocfg.result["power"][0] + ocfg.result["power"][1]).mean()           # mean total power PP + QQ
    err = np.abs(got - want).max() / scale
    print("max |P - P_oracle| / mean(PP+QQ) =", err)
    assert err <= pu.POWER_RTOL, err


@pytest.mark.parametrize("nchan,want", [(32, "frbch_k2_lane<1,2>"), (64, "frbch_k2_lane<2,2>")])
def test_few_channels_run_on_the_lane_kernel(hip_lib, nchan, want):
    from frb_baseband_amd import channeliser as ch, synth
    raw = synth.make_vdif(0.02, bw_mhz=16.0, nchan=nchan)
    with ch.Channeliser(pu.lib_cfg(hip_lib, 16.0, nchan, 0.02), hip_lib) as c:
        c.set_profiling(True)
        c.channelise_bytes(raw)
        names = {k for k, v in c.get_timing().items() if v["launches"]}
    assert want in names and any(n.startswith("frbch_k1_wave<1") for n in names), names
