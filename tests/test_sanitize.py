"""AddressSanitizer + UBSan over the host-side code (decoder, layout, path compression, packer) on
good and on corrupted inputs.  GPU sanitizers are not available on the pool; this is the CPU build."""
import gzip
import os
import random
import shutil
import subprocess

import pytest

from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cammiq_amd", "csrc")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("san") / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-o", out, os.path.join(ROOT, "tests", "cpp", "host_sanitize.cpp")] + \
          [os.path.join(SRC, f) for f in ("cq_decode.cpp", "cq_layout.cpp", "cq_pack.cpp")] + ["-lpthread"]
    subprocess.check_call(cmd)
    return out


DECODERS = {"serial": {"CAMMIQ_DECODE_THREADS": "1"},
            "chunked": {"CAMMIQ_DECODE_THREADS": "4", "CAMMIQ_DECODE_STEP": "3"},   # serial shape scan + parallel chunks of 3 buckets
            "scan": {"CAMMIQ_DECODE_THREADS": "4", "CAMMIQ_DECODE_SEG": "2"}}       # shape scan on all cores, segments of 2 bytes


def _run(driver, pu, pd, reads, decoder="serial"):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", **DECODERS[decoder])
    return subprocess.run([driver, pu, pd or "-", reads], capture_output=True, text=True, env=env)


@pytest.mark.parametrize("decoder", sorted(DECODERS))
@pytest.mark.parametrize("name", ["f_deep", "f_flat", "survey_F2"])
def test_host_code_is_clean_under_asan_ubsan(driver, tmp_path, name, decoder):
    g = golden(name)
    reads = tmp_path / "reads.txt"
    reads.write_bytes(b"\n".join(g["reads"][:2000]) + b"\nACGTNNNN\n\n" + bytes(range(1, 10)) + b"\n")
    r = _run(driver, g["pu"], g["pd"], str(reads), decoder)
    assert r.returncode == 0 and "\nok " in "\n" + r.stdout and "meta loaded" in r.stdout, r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_corrupted_indices_under_asan_ubsan(driver, tmp_path):
    g = golden("f_deep")
    files = ["index_u.bin1", "index_u.bin1.aux", "index_d.bin2", "index_d.bin2.aux"]
    for f in files:
        shutil.copy(os.path.join(g["dir"], f), tmp_path / f)
    reads = tmp_path / "reads.txt"
    reads.write_bytes(b"\n".join(g["reads"][:50]) + b"\n")
    rng = random.Random(3)
    good = {f: (tmp_path / f).read_bytes() for f in files}
    for trial in range(40):
        f = rng.choice(files)
        data = bytearray(good[f])
        if trial % 2:
            data = data[:rng.randrange(len(data))]
        else:
            for _ in range(rng.randrange(1, 6)):
                data[rng.randrange(len(data))] ^= 1 << rng.randrange(8)
        (tmp_path / f).write_bytes(bytes(data))
        outs = []
        for decoder in sorted(DECODERS):
            r = _run(driver, str(tmp_path / files[0]), str(tmp_path / files[2]), str(reads), decoder)
            assert r.returncode == 0, (trial, f, decoder, r.stdout, r.stderr[-2000:])
            assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
            outs.append(r.stdout)
        assert outs[0] == outs[1] == outs[2], "the chunked decoders and the serial one disagree on a corrupted file"
        (tmp_path / f).write_bytes(good[f])
