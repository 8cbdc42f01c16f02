"""The C-ABI digifil argv parser (shared by the CLI shim and the Python harness) against the
reference's golden command lines."""
import json
import os

import pytest

from frb_baseband_amd import digifil_args
from frb_baseband_amd.channeliser import InputError

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "harness_golden.json")))
POL_FROM_KW = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4}


@pytest.mark.parametrize("case", GOLD["run_digifil"], ids=lambda c: c["cmd"][-40:])
def test_parse_golden_commands(hip_lib, case):
    cfg, hdr, out = digifil_args.parse(case["cmd"], lib=hip_lib, read_hdr=False)
    kw = case["kwargs"]
    assert hdr == case["hdr"]
    assert out == case["returns"]
    assert cfg.nchan == kw["nchan"]
    assert cfg.nbit_out == kw["nbit"]
    assert cfg.pol_mode == POL_FROM_KW[kw["pol"]]
    assert cfg.start_s == float(kw["start"]) and cfg.total_s == float(kw["nsecs"])
    assert cfg.tscrunch == kw.get("tscrunch", 1)
    assert cfg.rescale_constant == 1                       # -c is always passed
    assert cfg.rescale_interval_s == (0.0 if kw.get("keepBP") else 10.0)
    assert cfg.dm == kw.get("dm", 0.0)                     # -D repeats: last wins
    if kw.get("coherent"):
        assert cfg.coherent == 1                           # -F repeats: last wins
    else:
        assert cfg.coherent == 0
        assert cfg.freq_res == (512 if kw["nchan"] <= 128 else 2 * kw["nchan"])


def test_parse_errors(hip_lib):
    for bad in ("digifil -o x.fil", "digifil x.hdr", "digifil -q -o x.fil x.hdr", "digifil -d2 -o x.fil x.hdr",
                "digifil -o x.fil x.hdr y.hdr", "digifil -o x.fil x.hdr -b"):
        with pytest.raises(InputError):
            digifil_args.parse(bad, lib=hip_lib, read_hdr=False)


def test_config_from_hdr(hip_lib, tmp_path):
    from frb_baseband_amd import process_vdif as pv
    fn = str(tmp_path / "a_IF1.vdif")
    hdr = pv.make_hdr("R3", 1340.49, fn, pol=2, usb=False, ra="01:58:00.7502", dec="65:43:00.3152", bw=32.0,
                      telescope="effelsberg")
    cfg, h, out = digifil_args.parse(f"digifil -cont -c -b8 -S0 -T1 -2 -D 0.0 -o {tmp_path}/o.fil {hdr} -threads 1 -d1 -F64:512",
                                     lib=hip_lib)
    assert cfg.bw_mhz == -32.0 and cfg.freq_mhz == 1340.49
    assert cfg.telescope == b"effelsberg" and cfg.source == b"R3"
    assert cfg.ra == b"01:58:00.7502" and cfg.dec == b"65:43:00.3152"
    assert cfg.datafile.decode() == fn
