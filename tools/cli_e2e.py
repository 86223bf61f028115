"""End-to-end `cammiq --query --read_cnts` on a configs[1]-sized input: 500-genome --unique index,
one FASTQ of 10 M x 100 bp reads in /dev/shm.  Prints the CLI's own stderr timings (index load,
query = FASTQ parse + pack + H2D + classify + D2H) and checks the TSV against the device API.
Run on the GPU box:  python tools/cli_e2e.py [--reads N]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cammiq_amd as cq
from cammiq_amd import bigsynth

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=500)
ap.add_argument("--reads", type=int, default=10_000_000)
a = ap.parse_args()
G, n, L = a.genomes, a.reads, 100
wdir = f"/dev/shm/cammiq_e2e_{os.getpid()}"
os.makedirs(wdir)
try:
    w = bigsynth.World(seed=2, n_genomes=G, genome_len=3_450_000, k=26, h=26, lmax=50, frac_deep=0.07, pair_share=0.0)
    pu = os.path.join(wdir, "index_u.bin1")
    w.write_index(pu, None)
    with open(os.path.join(wdir, "genome_map.out"), "w") as f:
        for i in range(1, G + 1):
            f.write(f"g{i}.fna\t{i}\t{2000 + i}\tsynthetic genome {i}\n")
    for name in ("genome_lengths.out", "unique_lmer_count_u.out", "unique_lmer_count_d.out"):
        with open(os.path.join(wdir, name), "w") as f:
            for i in range(1, G + 1):
                f.write(f"{i}\t3450000\n")
    bases, offs = w.reads(seed=1000, n=n, length=L)
    # FASTQ records "@r\n<100 bases>\n+\n<100 x I>\n" assembled as one byte matrix
    rec = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@r\n", np.uint8)
    rec[:, 3:3 + L] = bases.reshape(n, L)
    rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 6 + L:6 + 2 * L] = ord("I")
    rec[:, -1] = ord("\n")
    fq = os.path.join(wdir, "sample.fastq")
    rec.tofile(fq)
    del rec
    out = os.path.join(wdir, "out.tsv")
    def run_cli(extra, env_extra=None):
        t0 = time.time()
        r = subprocess.run([os.environ.get("CAMMIQ_CLI", os.path.join(ROOT, "cammiq_amd", "cammiq")), "--query", "--read_cnts", "-f",
                            os.path.join(wdir, "genome_map.out"), "-i", pu, "-q", fq, "-o", out] + extra,
                           capture_output=True, text=True, env=dict(os.environ, CAMMIQ_LOAD_TIMING="1", **(env_extra or {})))
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        times = {ln.split(":")[0].strip(): ln.split(":")[1].strip() for ln in r.stderr.splitlines()
                 if ln.startswith("Time for")}
        for ln in r.stderr.splitlines():
            if ln.startswith("[read_fastq]"):
                times["read_fastq"] = ln[len("[read_fastq]"):].strip()
        return wall, times
    wall, times = run_cli([])
    # the image cache FORCED (plain --image_cache skips it for a GPU handle whose table the device lays out: the faster load)
    run_cli(["--image_cache"], {"CAMMIQ_IMAGE_CACHE": "force"})                      # writes <index_u>.cqimg
    wall_cached, times_cached = run_cli(["--image_cache"], {"CAMMIQ_IMAGE_CACHE": "force"})
    tsv = open(out).read().splitlines()
    cli_counts = np.array(tsv[1].split("\t")[1:], dtype=np.int64)
    # the same reads through the library's host API
    ix = cq.Index(pu, None, device=0)
    ref = ix.query(bases, offs, G)
    want = np.asarray(ref["cnt_u"])[1:].astype(np.int64)
    assert np.array_equal(cli_counts, want), "CLI TSV differs from the library"
    ql = [v for k, v in times.items() if "query" in k]
    print(json.dumps({"reads": n, "genomes": G, "fastq_GB": round(os.path.getsize(fq) / 1e9, 3), "cli_wall_s": round(wall, 2),
                      "cli_stderr_times": times, "cli_wall_s_image_cache": round(wall_cached, 2),
                      "cli_stderr_times_image_cache": times_cached, "tsv_matches_library": True,
                      "query_Mreads_s": round(n / (float(ql[0].split()[0]) / 1e3) / 1e6, 1) if ql else None}))
finally:
    shutil.rmtree(wdir, ignore_errors=True)
